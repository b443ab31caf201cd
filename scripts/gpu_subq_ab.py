"""One work queue or eight sub-queues (RTW_OPT_SUB_QUEUES), blocks per grab (RTW_OPT_GRAB_BLOCKS) and tile order (RTW_OPT_TILE_ORDER) on every
config; kernel ms, best of 3.  (profiles/r02_order_ab.log holds three runs of this script with different triples, one of them with a
temporary mode "raster through the permutation table".)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtw_amd as R
out = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
cases = (("C1", R.SCENE_C1, R.SCENE_C1, 1, None), ("C2", R.SCENE_C2, R.SCENE_C2, 1, None), ("C3 full", R.SCENE_C2, R.SCENE_C5, 1, 0.0), ("C3 1/8", R.SCENE_C2, R.SCENE_C5, 8, 0.0),
         ("C4", R.SCENE_C4, R.SCENE_C4, 1, None), ("C4 1/8", R.SCENE_C4, R.SCENE_C4, 8, None), ("C5", R.SCENE_C5, R.SCENE_C5, 1, None), ("First frame", R.SCENE_FIRST_FRAME, R.SCENE_FIRST_FRAME, 1, None))
for name, sid, vid, parts, shutter in cases:
    scene = R.Scene.generate(sid); cam, p = R.default_view(vid)
    if shutter is not None: cam.shutter = shutter
    if parts > 1: p.row_block, p.part_index, p.part_count = 8, 3, parts
    with R.Renderer(0) as r:
        r.set_scene(scene, cam.time0, cam.time0 + cam.shutter)
        res = []
        for sq, grab, order in ((1, 2, 0), (0, 2, 0), (0, 1, 0), (0, 2, 2), (0, 2, 4), (0, 2, 3)):
            r.set_option(R.OPT_SUB_QUEUES, sq); r.set_option(R.OPT_GRAB_BLOCKS, grab); r.set_option(R.OPT_TILE_ORDER, order)
            best = min(r.render(cam, p, out=out.data_ptr())[1].kernel_ms for _ in range(3))
            res.append(f"{'one' if sq else 'eight'}/grab{grab}/order{order}: {best:.3f}")
        print(name, "  ".join(res), flush=True)
