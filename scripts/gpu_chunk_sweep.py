"""Work-unit length (RTW_OPT_CHUNK_LEN) on the full-size configs and on an eighth of the bench frame; kernel ms, best of 4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtw_amd as R
out = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
cases = [("C3", R.SCENE_C2, R.SCENE_C5, (1, 0)), ("C3/8 r0", R.SCENE_C2, R.SCENE_C5, (8, 0)), ("C3/8 r3", R.SCENE_C2, R.SCENE_C5, (8, 3)), ("C2", R.SCENE_C2, R.SCENE_C2, (1, 0)),
         ("C4", R.SCENE_C4, R.SCENE_C4, (1, 0)), ("C5", R.SCENE_C5, R.SCENE_C5, (1, 0))]
for name, sid, vid, part in cases:
    scene = R.Scene.generate(sid); cam, p = R.default_view(vid)
    if name.startswith("C3"): cam.shutter = 0.0
    p.row_block, p.part_count, p.part_index = 8, part[0], part[1]
    with R.Renderer(0) as r:
        r.set_scene(scene, cam.time0, cam.time0 + cam.shutter)
        res = {}
        for chunk in (4, 5, 6, 8):
            r.set_option(R.OPT_CHUNK_LEN, chunk)
            res[chunk] = min(r.render(cam, p, out=out.data_ptr())[1].kernel_ms for _ in range(4))
        print(f"{name:8s} " + "  ".join(f"chunk {c}: {v:8.3f} ms ({(res[4] / v - 1) * 100:+.2f}%)" for c, v in res.items()), flush=True)
