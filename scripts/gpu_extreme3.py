"""Exploration: odd quads / instances / media next to a field of spheres -- list walk and forced tree (GEOM builds) against the oracle."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import rtw_amd as R
from tests import oracle_binding as O
inf, nan = float("inf"), float("nan")
rng = np.random.default_rng(9)
mats = [R.SCATTER_M, R.METALLIC_M, R.GLASS_M]
def field():
    return [R.Sphere.with_albedo(rng.uniform(-4, 4, 3) + [0, 0, -8], float(rng.uniform(0.2, 0.6)), rng.uniform(0.2, 0.95, 3), mats[i % 3]) for i in range(70)]
r = R.Renderer(0)
vp = R.Viewport.new_from_res(96, 54, 4, 10, 1.0, vfov=70.0, lens_radius=0.0)
cam = vp.camera()
def run(name, scene, integ=R.INTEGRATOR_GRADIENT):
    p = vp.params(integ, R.SAMPLER_ROW)
    try:
        ref, st_ref = O.render(cam, scene, p, 8); r.set_scene(scene)
    except Exception as e:
        print(f"{name:44s}: oracle / set_scene raised {e}"); return
    res = []
    for label, walk_max, accel in (("list", 48, R.ACCEL_BRUTE), ("tree", 0, R.ACCEL_BVH)):
        r.set_option(R.OPT_LIST_WALK_MAX, walk_max); p.accel = accel
        try:
            img, st = r.render(cam, p)
            same = np.array_equal(img, ref, equal_nan=True)
            res.append(f"{label}: {'==' if same else 'DIFFERS (%d px)' % int((~np.isclose(img, ref, rtol=0, atol=0, equal_nan=True)).any(axis=2).sum())} segments {st.segments == st_ref.segments}")
        except Exception as e:
            res.append(f"{label}: raised {e}")
    r.set_option(R.OPT_LIST_WALK_MAX, 48)
    print(f"{name:44s}: " + " | ".join(res), flush=True)
def box(**kw):
    b = R.Instance.new_box((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0), (0.8, 0.8, 0.8), R.SCATTER_M)
    b.rotate(kw.get("rot", (0.0, 0.5, 0.0))); b.translate(kw.get("tr", (1.5, 0.0, -6.0)))
    if "density" in kw: b.const_density(kw["density"])
    return b
for integ, iname in ((R.INTEGRATOR_GRADIENT, "gradient"), (R.INTEGRATOR_BG_COLOR, "bg_color")):
    run(f"{iname}: ordinary quad + box", R.Scene(field(), background=(0.4, 0.5, 0.7), quads=[R.Quad.new((-3, -1, -9), (6, 0, 0), (0, 4, 0), R.METALLIC_M, (0.8, 0.8, 0.8))], instances=[box()]), integ)
    run(f"{iname}: degenerate quad (u x v = 0)", R.Scene(field(), background=(0.4, 0.5, 0.7), quads=[R.Quad.new((-3, -1, -9), (6, 0, 0), (3, 0, 0), R.SCATTER_M, (0.8, 0.8, 0.8))]), integ)
    run(f"{iname}: quad with a nan corner", R.Scene(field(), background=(0.4, 0.5, 0.7), quads=[R.Quad.new((nan, -1, -9), (6, 0, 0), (0, 4, 0), R.SCATTER_M, (0.8, 0.8, 0.8))]), integ)
    run(f"{iname}: quad at 1e20", R.Scene(field(), background=(0.4, 0.5, 0.7), quads=[R.Quad.new((1e20, -1, -9), (6, 0, 0), (0, 4, 0), R.SCATTER_M, (0.8, 0.8, 0.8))]), integ)
    run(f"{iname}: huge quad 1e20 wide", R.Scene(field(), background=(0.4, 0.5, 0.7), quads=[R.Quad.new((-5e19, -5e19, -12), (1e20, 0, 0), (0, 1e20, 0), R.SCATTER_M, (0.8, 0.8, 0.8))]), integ)
    run(f"{iname}: box rotated by nan", R.Scene(field(), background=(0.4, 0.5, 0.7), instances=[box(rot=(0.0, nan, 0.0))]), integ)
    run(f"{iname}: box translated to inf", R.Scene(field(), background=(0.4, 0.5, 0.7), instances=[box(tr=(inf, 0.0, -6.0))]), integ)
    run(f"{iname}: box translated to 1e25", R.Scene(field(), background=(0.4, 0.5, 0.7), instances=[box(tr=(1e25, 0.0, -6.0))]), integ)
    for d in (0.0, -1.0, nan, inf, 1e-30, 1e30):
        run(f"{iname}: smoke of density {d}", R.Scene(field(), background=(0.4, 0.5, 0.7), instances=[box(density=d)]), integ)
