"""Lane census of the sub-blocks of a SHADE step (-DRTW_CENSUS build: RTW_HIP_LIB=.../librtw_hip_census.so RTW_CENSUS_DUMP=1).
Renders one frame of the chosen config (default: the bench frame) and lets the library print the table to stderr."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import rtw_amd as R
cfg = os.environ.get("CONFIG", "c3")
if cfg == "c3":
    scene = R.Scene.generate(R.SCENE_C2, 42)
    cam, p = R.default_view(R.SCENE_C5); cam.shutter = 0.0
else:
    which = {"c2": R.SCENE_C2, "c4": R.SCENE_C4, "c5": R.SCENE_C5}[cfg]
    scene = R.Scene.generate(which, 42)
    cam, p = R.default_view(which)
r = R.Renderer(0); r.set_scene(scene, cam.time0, cam.time0 + cam.shutter)
out = torch.zeros((p.height, p.width, 3), dtype=torch.float32, device="cuda:0")
sys.stderr.write(f"--- census of config {cfg}\n"); sys.stderr.flush()
_, st = r.render(cam, p, out=out.data_ptr())
sys.stderr.write(f"    kernel {st.kernel_ms:.3f} ms, {st.segments} segments, {st.camera_rays} camera rays; scheduler census: " +
                 " ".join(f"{n} {st.phase_lanes[k] / max(1, 64 * st.phase_steps[k]):.3f}/{st.phase_steps[k]}" for k, n in enumerate(("traverse", "leaf", "shade"))) + "\n")
