"""Where does a small frame's time go?  C1 (400x225x10, 3 spheres), First frame (400x400x100, 7 spheres), C2 and the bench frame split 8
ways: kernel ms (HIP events) and host wall ms of rtw_ctx_render, best of 7, under chunk length / resident workgroups / list-walk options."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtw_amd as R

def best(r, cam, p, n=7):
    ks, ts, st = [], [], None
    for _ in range(n):
        _, st = r.render(cam, p, out=OUT)
        ks.append(st.kernel_ms); ts.append(st.total_ms)
    return min(ks), min(ts), st

import torch
OUT = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0").data_ptr()
cases = [("C1", R.SCENE_C1, R.SCENE_C1, None), ("first_frame", R.SCENE_FIRST_FRAME, R.SCENE_FIRST_FRAME, None),
         ("C2", R.SCENE_C2, R.SCENE_C2, None), ("C3/8 (rank 0 of 8)", R.SCENE_C2, R.SCENE_C5, (8, 0, 8))]
for name, sid, vid, part in cases:
    scene = R.Scene.generate(sid)
    cam, p = R.default_view(vid)
    if vid == R.SCENE_C5: cam.shutter = 0.0
    if part: p.row_block, p.part_index, p.part_count = part
    with R.Renderer(0) as r:
        r.set_scene(scene, cam.time0, cam.time0 + cam.shutter)
        for accel, lw in ((R.ACCEL_BRUTE, 8), (R.ACCEL_BVH, 0)):
            if sid == R.SCENE_C2 and accel == R.ACCEL_BRUTE: continue
            p.accel = accel
            r.set_option(R.OPT_LIST_WALK_MAX, lw)
            for chunk in (1, 2, 4, 8):
                for bpc in (0, 1, 2, 4):
                    r.set_option(R.OPT_CHUNK_LEN, chunk); r.set_option(R.OPT_BLOCKS_PER_CU, bpc)
                    k, t, st = best(r, cam, p)
                    print(f"{name:20s} accel {accel} chunk {chunk} blocks/CU {bpc or 'auto'}: kernel {k:7.3f} ms  wall {t:7.3f} ms  {st.segments / k / 1e6:8.2f} Gseg/s", flush=True)
