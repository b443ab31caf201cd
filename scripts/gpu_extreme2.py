"""Exploration: odd t-ranges and odd material scalars -- list walk, BVH request and forced tree against the oracle."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import rtw_amd as R
from tests import oracle_binding as O
inf, nan = float("inf"), float("nan")
rng = np.random.default_rng(7)
mats = [R.SCATTER_M, R.METALLIC_M, R.GLASS_M, R.FUZZY3_M]
def spheres(odd_mat=None):
    sp = [R.Sphere.with_albedo((0, -100.5, -8), 100.0, (0.5, 0.5, 0.5), R.SCATTER_M)]
    sp += [R.Sphere.with_albedo(rng.uniform(-4, 4, 3) + [0, 0, -8], float(rng.uniform(0.2, 0.6)), rng.uniform(0.2, 0.95, 3), mats[i % 4]) for i in range(70)]
    if odd_mat is not None:
        for i in range(1, 70, 3): sp[i] = R.Sphere.with_albedo(tuple(sp[i].pod.center), sp[i].pod.radius, odd_mat[1], odd_mat[0])
    return sp
r = R.Renderer(0)
vp = R.Viewport.new_from_res(96, 54, 4, 12, 1.0, vfov=70.0, lens_radius=0.02)
cam = vp.camera()
def run(name, scene, p):
    try:
        ref, st_ref = O.render(cam, scene, p, 8); r.set_scene(scene)
    except Exception as e:
        print(f"{name:40s}: oracle / set_scene raised {e}"); return
    res = []
    for label, walk_max, accel in (("list", 48, R.ACCEL_BRUTE), ("tree", 0, R.ACCEL_BVH)):
        r.set_option(R.OPT_LIST_WALK_MAX, walk_max); p.accel = accel
        try:
            img, st = r.render(cam, p)
            same = np.array_equal(img, ref, equal_nan=True)
            res.append(f"{label}: {'==' if same else 'DIFFERS (%d px, max %.3g)' % (int((~np.isclose(img, ref, rtol=0, atol=0, equal_nan=True)).any(axis=2).sum()), np.nanmax(np.abs(np.nan_to_num(img) - np.nan_to_num(ref))))} segments {st.segments == st_ref.segments}")
        except Exception as e:
            res.append(f"{label}: raised {e}")
    r.set_option(R.OPT_LIST_WALK_MAX, 48)
    print(f"{name:40s}: " + " | ".join(res), flush=True)
base = R.Scene(spheres())
for mint, maxt in ((0.001, inf), (0.001, nan), (0.001, 3e38), (0.001, -1.0), (0.001, 5.0), (0.0, 1e5), (-1.0, 1e5), (nan, 1e5), (inf, 1e5), (1e-30, 1e5), (7.0, 9.0)):
    p = vp.params(R.INTEGRATOR_GRADIENT, R.SAMPLER_ROW); p.mint, p.maxt = mint, maxt
    run(f"mint {mint} maxt {maxt}", base, p)
p = vp.params(R.INTEGRATOR_GRADIENT, R.SAMPLER_ROW)
for name, m, alb in (("ir 0 glass", (1.0, 1.0, 0.0), (1, 1, 1)), ("ir nan glass", (1.0, 1.0, nan), (1, 1, 1)), ("ir inf glass", (1.0, 1.0, inf), (1, 1, 1)), ("ir negative glass", (1.0, 1.0, -1.5), (1, 1, 1)),
                     ("metallicness nan", (nan, 0.0, 1.0), (0.8, 0.8, 0.8)), ("metallicness 2", (2.0, 0.0, 1.0), (0.8, 0.8, 0.8)), ("metallicness -1", (-1.0, 0.0, 1.0), (0.8, 0.8, 0.8)),
                     ("opacity nan", (0.5, nan, 1.5), (0.8, 0.8, 0.8)), ("opacity -1", (0.5, -1.0, 1.5), (0.8, 0.8, 0.8)), ("albedo nan", R.SCATTER_M, (nan, 0.5, 0.5)),
                     ("albedo inf", R.SCATTER_M, (inf, 0.5, 0.5)), ("albedo negative", R.FUZZY3_M, (-0.5, 0.5, 0.5)), ("albedo 1e30", R.METALLIC_M, (1e30, 1e30, 1e30))):
    run(name, R.Scene(spheres((m, alb))), p)
for d in (0, 1, 2, 200):
    p = vp.params(R.INTEGRATOR_GRADIENT, R.SAMPLER_ROW); p.depth = d
    run(f"depth {d}", base, p)
