"""Exploration: spheres with extreme / non-finite coordinates among ordinary ones -- tree (forced) vs list walk vs oracle."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import rtw_amd as R
from tests import oracle_binding as O
rng = np.random.default_rng(5)
mats = [R.SCATTER_M, R.METALLIC_M, R.GLASS_M]
def base():
    return [R.Sphere.with_albedo(rng.uniform(-4, 4, 3) + [0, 0, -8], float(rng.uniform(0.2, 0.6)), rng.uniform(0.2, 0.95, 3), mats[i % 3]) for i in range(60)]
inf, nan = float("inf"), float("nan")
cases = {"far 1e15": ((1e15, 0, -8), 1.0), "far 1e17": ((1e17, 0, -8), 1.0), "far 4e17": ((4e17, 3e17, -8), 1.0), "far 6e17 (list walk)": ((6e17, 0, -8), 1.0),
         "far 1e17 radius 1e17 (grazing the field)": ((1e17, 0, -8), 1e17), "far 1e12 radius 1e12 - 3": ((1e12, 0, -8), 1e12),
         "far 1e19": ((1e19, 0, -8), 1.0), "far 1e30": ((1e30, 0, -8), 1.0), "giant radius 1e20": ((0, 0, -8), 1e20), "radius 1e10 around everything": ((0, 0, -8), 1e10),
         "centre inf": ((inf, 0, -8), 1.0), "centre nan": ((nan, 0, -8), 1.0), "radius nan": ((0, 1, -8), nan), "radius inf": ((0, 1, -8), inf),
         "radius 0": ((0, 1, -6), 0.0), "radius negative": ((0.5, 0.5, -6), -0.7), "centre -1e25 (behind)": ((0, 0, 1e25), 3.0)}
r = R.Renderer(0)
vp = R.Viewport.new_from_res(96, 54, 4, 8, 1.0, vfov=70.0, lens_radius=0.0)
cam = vp.camera(); p = vp.params(R.INTEGRATOR_GRADIENT, R.SAMPLER_ROW)
for name, (c, rad) in cases.items():
    for pos in (0, 30, 60):                    # the odd sphere first / in the middle / last in list order
        sp = base(); sp.insert(pos, R.Sphere.with_albedo(c, rad, (0.9, 0.2, 0.2), R.SCATTER_M))
        scene = R.Scene(sp)
        try:
            ref, st_ref = O.render(cam, scene, p, 8)
            r.set_scene(scene)
        except Exception as e:
            print(f"{name:32s} pos {pos:2d}: set_scene / oracle raised {e}"); continue
        res = []
        for walk_max, accel in ((48, R.ACCEL_BRUTE), (0, R.ACCEL_BVH)):
            r.set_option(R.OPT_LIST_WALK_MAX, walk_max); p.accel = accel
            try:
                img, st = r.render(cam, p)
                res.append(f"{'list' if accel == R.ACCEL_BRUTE else 'tree'}: {'==' if np.array_equal(img, ref, equal_nan=True) else 'DIFFERS (%d px)' % int((~np.isclose(img, ref, rtol=0, atol=0, equal_nan=True)).any(axis=2).sum())} oracle, segments {st.segments == st_ref.segments}")
            except Exception as e:
                res.append(f"render raised {e}")
        print(f"{name:32s} pos {pos:2d}: " + " | ".join(res), flush=True)
