"""Blocks per queue atomic (RTW_OPT_GRAB_BLOCKS: 1 = single 64-item blocks, N = guided grabs of at most N blocks, 0 = at most one tile's
blocks) on the bench frame, an eighth of it, C2, C4 and C5.  Kernel ms, best of 3, and the TRAVERSE lane efficiency."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtw_amd as R
out = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
cases = (("C3 full", R.SCENE_C2, R.SCENE_C5, 1, 0.0), ("C3 1/8", R.SCENE_C2, R.SCENE_C5, 8, 0.0), ("C2", R.SCENE_C2, R.SCENE_C2, 1, None),
         ("C4", R.SCENE_C4, R.SCENE_C4, 1, None), ("C5", R.SCENE_C5, R.SCENE_C5, 1, None))
for name, sid, vid, parts, shutter in cases:
    scene = R.Scene.generate(sid); cam, p = R.default_view(vid)
    if shutter is not None: cam.shutter = shutter
    if parts > 1: p.row_block, p.part_index, p.part_count = 8, 3, parts
    with R.Renderer(0) as r:
        r.set_scene(scene, cam.time0, cam.time0 + cam.shutter)
        res = []
        for g in (1, 2, 4, 8, 16, 0):
            r.set_option(R.OPT_GRAB_BLOCKS, g)
            best = None
            for _ in range(3):
                _, st = r.render(cam, p, out=out.data_ptr())
                if best is None or st.kernel_ms < best.kernel_ms: best = st
            res.append(f"{g if g else 'tile'}: {best.kernel_ms:.3f} ({best.phase_lanes[0] / max(1, best.phase_steps[0]) / 64:.3f})")
        print(name, "  ".join(res), flush=True)
