"""End of a launch: tile order (RTW_OPT_TILE_ORDER 0 raster / 5 raster with sky last / 2 cost order) x guided tail units (RTW_OPT_TAIL_UNITS 0 / 2)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import rtw_amd as R
def bench_frame():
    scene = R.Scene.generate(R.SCENE_C2, 42)
    cam, p = R.default_view(R.SCENE_C5); cam.shutter = 0.0
    return scene, cam, p
cases = []
sc, cam, p = bench_frame(); cases.append(("C3 bench frame", sc, cam, p, (0, 1)))
for idx in (0, 3):
    sc, cam, p = bench_frame(); cases.append((f"an eighth of it (rank {idx} of 8)", sc, cam, p, (idx, 8)))
for name, which in (("C2", R.SCENE_C2), ("C4", R.SCENE_C4), ("C5", R.SCENE_C5)):
    sc = R.Scene.generate(which, 42); cam, p = R.default_view(which); cases.append((name, sc, cam, p, (0, 1)))
r = R.Renderer(0)
for name, sc, cam, p, (idx, cnt) in cases:
    r.set_scene(sc, cam.time0, cam.time0 + cam.shutter)
    p.accel = R.ACCEL_BVH
    if cnt > 1: p.row_block, p.part_index, p.part_count = 8, idx, cnt
    rows = R.lib().rtw_part_rows(p.height, p.row_block, p.part_index, p.part_count)
    out = torch.zeros((rows, p.width, 3), dtype=torch.float32, device="cuda:0")
    ref = None; line = []
    for order in (0, 5, 2):
        for k in (0.0, 2.0):
            r.set_option(R.OPT_TILE_ORDER, order); r.set_option(R.OPT_TAIL_UNITS, k)
            r.render(cam, p, out=out.data_ptr())
            best = min(r.render(cam, p, out=out.data_ptr())[1].kernel_ms for _ in range(3))
            img = out.cpu().numpy()
            if ref is None: ref = img
            line.append(f"order {order} k={k:g}: {best:.3f}" + ("" if np.array_equal(img, ref) else " IMAGE DIFFERS"))
    print(f"{name:34s} " + "   ".join(line), flush=True)
