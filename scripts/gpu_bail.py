"""The end of a launch, consolidated (RTW_OPT_HAND_LANES): kernel time of the bench frame, of an eighth of it (one rank's share of N = 8) and of C2 for
several thresholds, with an md5 of every image (the option must not change a bit).  Library: RTW_HIP_LIB."""
import os, sys, hashlib
sys.path.insert(0, os.getcwd())
import torch
import rtw_amd as R
r = R.Renderer(0)
if os.environ.get("TAIL"): r.set_option(R.OPT_TAIL_UNITS, float(os.environ["TAIL"]))
vals = [int(x) for x in os.environ.get("BAIL", "0,4,8,16,32,64").split(",")]
def frame(name, scene_id, view_id, shutter, part):
    sc = R.Scene.generate(scene_id, 42); cam, p = R.default_view(view_id)
    if shutter is not None: cam.shutter = shutter
    if part: p.row_block, p.part_index, p.part_count = 8, part[0], part[1]
    r.set_scene(sc, cam.time0, cam.time0 + cam.shutter)
    rows = p.height if not part else R.part_rows(p.height, 8, part[0], part[1]) if hasattr(R, "part_rows") else p.height
    out = torch.zeros((p.height, p.width, 3), dtype=torch.float32, device="cuda:0")
    line = f"{name:34s}"
    for b in vals:
        if hasattr(R, "OPT_HAND_LANES"):
            try: r.set_option(R.OPT_HAND_LANES, b)
            except Exception: pass
        out.zero_()
        r.render(cam, p, out=out.data_ptr())
        best = min((r.render(cam, p, out=out.data_ptr())[1] for _ in range(4)), key=lambda st: st.kernel_ms)
        line += f"  bail {b:2d}: {best.kernel_ms:7.3f} ms {hashlib.md5(out.cpu().numpy().tobytes()).hexdigest()[:6]}"
    print(line, flush=True)
frame("C3 bench frame", R.SCENE_C2, R.SCENE_C5, 0.0, None)
frame("an eighth of it (rank 0 of 8)", R.SCENE_C2, R.SCENE_C5, 0.0, (0, 8))
frame("an eighth of it (rank 3 of 8)", R.SCENE_C2, R.SCENE_C5, 0.0, (3, 8))
frame("C2", R.SCENE_C2, R.SCENE_C2, None, None)
frame("C4", R.SCENE_C4, R.SCENE_C4, None, None)
frame("C5", R.SCENE_C5, R.SCENE_C5, None, None)
