"""Raster vs scattered tile order: the bench frame whole and split 8 ways (rank 0, 3, 7), C2; kernel ms best of 5; image must not change."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtw_amd as R
out = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
ref = torch.zeros_like(out)
for name, sid, vid in (("C3", R.SCENE_C2, R.SCENE_C5), ("C2", R.SCENE_C2, R.SCENE_C2), ("C4", R.SCENE_C4, R.SCENE_C4), ("C5", R.SCENE_C5, R.SCENE_C5)):
    scene = R.Scene.generate(sid)
    cam, p = R.default_view(vid)
    if name == "C3": cam.shutter = 0.0
    with R.Renderer(0) as r:
        r.set_scene(scene, cam.time0, cam.time0 + cam.shutter)
        for part in ((1, 0), (8, 0), (8, 3), (8, 7)):
            if name != "C3" and part[0] != 1: continue
            p.row_block, p.part_count, p.part_index = 8, part[0], part[1]
            res = {}
            rows = R.lib().rtw_part_rows(p.height, 8, part[1], part[0])
            same = True
            for order in (0, 1, 2, 3):
                r.set_option(R.OPT_TILE_ORDER, order)
                dst = ref if order == 0 else out
                best = min(r.render(cam, p, out=dst.data_ptr())[1].kernel_ms for _ in range(4))
                res[order] = best
                if order:
                    same &= bool(torch.equal(ref.view(-1)[:rows * p.width * 3], out.view(-1)[:rows * p.width * 3]))
            print(f"{name} part {part[1]}/{part[0]}: raster {res[0]:8.3f} ms | " + "  ".join(f"{n} {(res[0] / res[k] - 1) * 100:+.2f}%" for k, n in ((1, "groups8"), (2, "groups8+cheap-last"), (3, "reverse"))) + f" | images identical: {same}", flush=True)
