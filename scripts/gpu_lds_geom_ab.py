"""Does keeping the spheres' {centre, r^2} in LDS pay for mid-sized scenes?  N random spheres over a ground sphere, 800x450x32 spp, depth 20,
tree forced; kernel ms best of 5 with RTW_OPT_LDS_GEOM = 0 (global, L2-resident) and 1 (LDS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rtw_amd as R
out = torch.zeros((450, 800, 3), dtype=torch.float32, device="cuda:0")
vp = R.Viewport.new_from_res(800, 450, 32, 20, 2.0, vfov=40.0, origin=(0.0, 1.5, 6.0), direction=(0.0, -0.2, -1.0))
cam, p = vp.camera(), vp.params(R.INTEGRATOR_GRADIENT, R.SAMPLER_ROW)
p.accel = R.ACCEL_BVH
with R.Renderer(0) as r:
    r.set_option(R.OPT_LIST_WALK_MAX, 0)
    for n in (16, 48, 64, 96, 128, 192, 256, 320, 400):
        rng = np.random.default_rng(n)
        mats = [R.SCATTER_M, R.METALLIC_M, R.GLASS_M, R.FUZZY3_M]
        sp = [R.Sphere.with_albedo((0, -1000, 0), 1000.0, (0.5, 0.5, 0.5))]
        sp += [R.Sphere.with_albedo((float(rng.uniform(-6, 6)), float(rng.uniform(0.2, 1.2)), float(rng.uniform(-6, 2))), float(rng.uniform(0.1, 0.3)),
                                    tuple(rng.uniform(0.3, 0.9, 3)), mats[i % 4]) for i in range(n)]
        r.set_scene(R.Scene(sp))
        res = {}
        for g in (0, 1, 0, 1):
            r.set_option(R.OPT_LDS_GEOM, g)
            best = min(r.render(cam, p, out=out.data_ptr())[1].kernel_ms for _ in range(5))
            res[g] = min(best, res.get(g, 1e9))
        print(f"{n + 1:4d} spheres: global {res[0]:7.3f} ms   LDS {res[1]:7.3f} ms   ({(res[0] / res[1] - 1) * 100:+.1f} % for LDS)", flush=True)
