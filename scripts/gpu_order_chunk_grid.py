"""Tile order x unit length on the full-size configs and an eighth of the bench frame (one box, kernel ms best of 4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtw_amd as R
out = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
cases = [("C3", R.SCENE_C2, R.SCENE_C5, (1, 0)), ("C3/8 r0", R.SCENE_C2, R.SCENE_C5, (8, 0)), ("C3/8 r3", R.SCENE_C2, R.SCENE_C5, (8, 3)), ("C3/2 r1", R.SCENE_C2, R.SCENE_C5, (2, 1)),
         ("C2", R.SCENE_C2, R.SCENE_C2, (1, 0)), ("C4", R.SCENE_C4, R.SCENE_C4, (1, 0)), ("C5", R.SCENE_C5, R.SCENE_C5, (1, 0))]
for name, sid, vid, part in cases:
    scene = R.Scene.generate(sid); cam, p = R.default_view(vid)
    if name.startswith("C3"): cam.shutter = 0.0
    p.row_block, p.part_count, p.part_index = 8, part[0], part[1]
    with R.Renderer(0) as r:
        r.set_scene(scene, cam.time0, cam.time0 + cam.shutter)
        res = {}
        for order in (0, 2, 3):
            for chunk in (4, 6, 8):
                r.set_option(R.OPT_TILE_ORDER, order); r.set_option(R.OPT_CHUNK_LEN, chunk)
                res[(order, chunk)] = min(r.render(cam, p, out=out.data_ptr())[1].kernel_ms for _ in range(4))
        best = min(res.values())
        print(f"{name:8s} " + "  ".join(f"o{o}c{c} {v:7.3f}{'*' if v == best else ' '}" for (o, c), v in res.items()), flush=True)
