"""Render one of the generator scenes a few times (for rocprofv3): python3 scripts/run_scene.py presentation|quads|firstframe|c1 [reps] [accel]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtw_amd as R
name = sys.argv[1] if len(sys.argv) > 1 else "presentation"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
which = {"presentation": R.SCENE_PRESENTATION, "quads": R.SCENE_QUAD_TEST, "firstframe": R.SCENE_FIRST_FRAME, "c1": R.SCENE_C1}[name]
sc = R.Scene.generate_geom(which) if which in (R.SCENE_QUAD_TEST, R.SCENE_PRESENTATION) else R.Scene.generate(which)
cam, p = R.default_view(which)
if len(sys.argv) > 3: p.accel = R.ACCEL_BRUTE if sys.argv[3] == "brute" else R.ACCEL_BVH
out = torch.zeros((p.height, p.width, 3), dtype=torch.float32, device="cuda:0")
with R.Renderer(0) as r:
    r.set_scene(sc, cam.time0, cam.time0 + cam.shutter)
    for _ in range(reps):
        _, st = r.render(cam, p, out=out.data_ptr())
    print(f"{name}: {st.kernel_ms:.2f} ms, {st.segments / st.kernel_ms / 1e6:.2f} Gseg/s, quad tests/seg {st.quad_tests / st.segments:.2f}, sphere tests/seg {st.sphere_tests / st.segments:.2f}")
