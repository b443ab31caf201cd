"""At how many spheres does the BVH kernel overtake the list walk?  N random spheres over a ground sphere, 800x450x32 spp, depth 20;
kernel ms best of 5 for RTW_ACCEL_BRUTE and RTW_ACCEL_BVH (tree forced: RTW_OPT_LIST_WALK_MAX = 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rtw_amd as R
out = torch.zeros((450, 800, 3), dtype=torch.float32, device="cuda:0")
vp = R.Viewport.new_from_res(800, 450, 32, 20, 2.0, vfov=40.0, origin=(0.0, 1.5, 6.0), direction=(0.0, -0.2, -1.0))
cam, p = vp.camera(), vp.params(R.INTEGRATOR_GRADIENT, R.SAMPLER_ROW)
with R.Renderer(0) as r:
    r.set_option(R.OPT_LIST_WALK_MAX, 0)
    for n in (1, 2, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128):
        rng = np.random.default_rng(n)
        mats = [R.SCATTER_M, R.METALLIC_M, R.GLASS_M, R.FUZZY3_M]
        sp = [R.Sphere.with_albedo((0, -1000, 0), 1000.0, (0.5, 0.5, 0.5))]
        sp += [R.Sphere.with_albedo((float(rng.uniform(-4, 4)), float(rng.uniform(0.2, 1.2)), float(rng.uniform(-4, 2))), float(rng.uniform(0.15, 0.45)),
                                    tuple(rng.uniform(0.3, 0.9, 3)), mats[i % 4]) for i in range(n)]
        r.set_scene(R.Scene(sp))
        res = {}
        for accel in (R.ACCEL_BRUTE, R.ACCEL_BVH):
            p.accel = accel
            best = min(r.render(cam, p, out=out.data_ptr())[1].kernel_ms for _ in range(5))
            res[accel] = best
        print(f"{n + 1:4d} spheres: list walk {res[R.ACCEL_BRUTE]:7.3f} ms   tree {res[R.ACCEL_BVH]:7.3f} ms   -> {'list' if res[R.ACCEL_BRUTE] < res[R.ACCEL_BVH] else 'tree'}", flush=True)
