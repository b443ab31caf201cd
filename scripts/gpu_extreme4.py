"""Exploration: odd textures, odd times with moving spheres, Rust2's sampler with odd sample counts -- list walk and forced tree against the oracle."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import rtw_amd as R
from tests import oracle_binding as O
inf, nan = float("inf"), float("nan")
rng = np.random.default_rng(11)
mats = [R.SCATTER_M, R.METALLIC_M, R.GLASS_M]
r = R.Renderer(0)
def run(name, scene, cam, p, tol_px=0):
    try:
        ref, st_ref = O.render(cam, scene, p, 8); r.set_scene(scene, cam.time0, cam.time0 + cam.shutter)
    except Exception as e:
        print(f"{name:46s}: oracle / set_scene raised {str(e)[:80]}"); return
    res = []
    for label, walk_max, accel in (("list", 48, R.ACCEL_BRUTE), ("tree", 0, R.ACCEL_BVH)):
        r.set_option(R.OPT_LIST_WALK_MAX, walk_max); p.accel = accel
        try:
            img, st = r.render(cam, p)
            bad = int((~np.isclose(img, ref, rtol=0, atol=0, equal_nan=True)).any(axis=2).sum())
            res.append(f"{label}: {'==' if bad == 0 else 'differs in %d px' % bad} segments {st.segments == st_ref.segments}")
        except Exception as e:
            res.append(f"{label}: raised {str(e)[:60]}")
    r.set_option(R.OPT_LIST_WALK_MAX, 48)
    print(f"{name:46s}: " + " | ".join(res), flush=True)
vp = R.Viewport.new_from_res(96, 54, 4, 10, 1.0, vfov=70.0, lens_radius=0.01)
# textures
for name, tex in (("1x1", np.float32([[[0.3, 0.6, 0.9]]])), ("1x7", rng.uniform(0.1, 0.9, (1, 7, 3)).astype(np.float32)), ("7x1", rng.uniform(0.1, 0.9, (7, 1, 3)).astype(np.float32)),
                  ("2048x2", rng.uniform(0.1, 0.9, (2, 2048, 3)).astype(np.float32)), ("nan texels", np.full((3, 3, 3), nan, np.float32)), ("inf texels", np.full((3, 3, 3), inf, np.float32)),
                  ("negative texels", -rng.uniform(0.1, 0.9, (4, 4, 3)).astype(np.float32))):
    sp = [R.Sphere.new_with_texture(tuple(rng.uniform(-4, 4, 3) + [0, 0, -8]), float(rng.uniform(0.3, 0.7)), None, mats[i % 2], 0) if i % 2 == 0 else
          R.Sphere.with_albedo(rng.uniform(-4, 4, 3) + [0, 0, -8], float(rng.uniform(0.2, 0.6)), rng.uniform(0.2, 0.95, 3), mats[i % 3]) for i in range(70)]
    cam = vp.camera(); p = vp.params(R.INTEGRATOR_GRADIENT, R.SAMPLER_ROW)
    run(f"texture {name}", R.Scene(sp, textures=[tex]), cam, p)
# times
for name, t0, sh in (("shutter nan", 0.0, nan), ("time0 inf", inf, 0.0), ("time0 nan", nan, 0.02), ("shutter negative", 0.5, -0.03), ("shutter 1e30", 0.0, 1e30), ("time0 -1e10", -1e10, 0.0)):
    sp = [R.Sphere.with_albedo(rng.uniform(-4, 4, 3) + [0, 0, -8], float(rng.uniform(0.2, 0.6)), rng.uniform(0.2, 0.95, 3), mats[i % 3], velocity=tuple(rng.uniform(-3, 3, 3)) if i % 3 == 0 else (0, 0, 0)) for i in range(70)]
    cam = vp.camera(); cam.time0, cam.shutter = t0, sh
    p = vp.params(R.INTEGRATOR_GRADIENT, R.SAMPLER_ROW)
    run(f"moving: {name}", R.Scene(sp), cam, p)
# Rust2's model, odd sample counts
from tests.test_oracle_golden import rust2_view
for n in (1, 2, 3, 5, 8, 10):
    scene, cam, p = rust2_view(64, 36, n, 6)
    sp = [R.Sphere.with_albedo(rng.uniform(-3, 3, 3) + [0, 0, -5], float(rng.uniform(0.2, 0.5)), rng.uniform(0.2, 0.95, 3), mats[i % 3]) for i in range(60)]
    run(f"Rust2 model, samples {n}", R.Scene(sp, background=(0.6, 0.7, 0.9)), cam, p)
