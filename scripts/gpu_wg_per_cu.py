import os, sys
sys.path.insert(0, os.getcwd())
import torch
import rtw_amd as R
out = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
scene = R.Scene.generate(R.SCENE_C2)
with R.Renderer(0) as r:
    r.set_scene(scene)
    for parts in (1, 2, 4, 8):
        cam, p = R.default_view(R.SCENE_C5); cam.shutter = 0.0
        if parts > 1: p.row_block, p.part_index, p.part_count = 8, parts // 2, parts
        res = []
        for b in (0, 6, 7):
            r.set_option(R.OPT_BLOCKS_PER_CU, b)
            res.append(f"{'auto' if b == 0 else b}: {min(r.render(cam, p, out=out.data_ptr())[1].kernel_ms for _ in range(4)):.3f}")
        print(f"C3 1/{parts}", "  ".join(res), flush=True)
    scene2 = R.Scene.generate(R.SCENE_C2); cam, p = R.default_view(R.SCENE_C2)
    r.set_scene(scene2)
    res = []
    for b in (0, 6, 7):
        r.set_option(R.OPT_BLOCKS_PER_CU, b)
        res.append(f"{'auto' if b == 0 else b}: {min(r.render(cam, p, out=out.data_ptr())[1].kernel_ms for _ in range(4)):.3f}")
    print("C2", "  ".join(res), flush=True)
