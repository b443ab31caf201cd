import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtw_amd as R
out = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
cases = [("C3", R.SCENE_C2, R.SCENE_C5, (1, 0)), ("C3/8 r0", R.SCENE_C2, R.SCENE_C5, (8, 0)), ("C3/8 r4", R.SCENE_C2, R.SCENE_C5, (8, 4)), ("C2", R.SCENE_C2, R.SCENE_C2, (1, 0)),
         ("C4", R.SCENE_C4, R.SCENE_C4, (1, 0)), ("C5", R.SCENE_C5, R.SCENE_C5, (1, 0)), ("C1", R.SCENE_C1, R.SCENE_C1, (1, 0)), ("first", R.SCENE_FIRST_FRAME, R.SCENE_FIRST_FRAME, (1, 0))]
res = []
for name, sid, vid, part in cases:
    scene = R.Scene.generate(sid); cam, p = R.default_view(vid)
    if name.startswith("C3"): cam.shutter = 0.0
    p.row_block, p.part_count, p.part_index = 8, part[0], part[1]
    with R.Renderer(0) as r:
        r.set_scene(scene, cam.time0, cam.time0 + cam.shutter)
        if name in ("C1", "first"): r.set_option(R.OPT_LIST_WALK_MAX, 0)      # (force the tree kernel: that is what changed)
        res.append(f"{name} {min(r.render(cam, p, out=out.data_ptr())[1].kernel_ms for _ in range(5)):.3f}")
print(os.environ.get("RTW_HIP_LIB", "default").split("/")[-1], "  ".join(res), flush=True)
