"""Unit length (samples per work unit) against launch size, after the kernel changes of late round 2: the bench frame, an eighth of it
(one rank of an 8-GPU job) and C2.  Kernel ms, best of 4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtw_amd as R
out = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
scene = R.Scene.generate(R.SCENE_C2)
with R.Renderer(0) as r:
    r.set_scene(scene)
    for name, vid, parts in (("C3 full", R.SCENE_C5, 1), ("C3 1/8", R.SCENE_C5, 8), ("C2", R.SCENE_C2, 1)):
        cam, p = R.default_view(vid); cam.shutter = 0.0
        if parts > 1: p.row_block, p.part_index, p.part_count = 8, 3, parts
        res = {}
        for chunk in (0, 4, 6, 8, 10, 12, 16):
            r.set_option(R.OPT_CHUNK_LEN, chunk)
            res[chunk] = min(r.render(cam, p, out=out.data_ptr())[1].kernel_ms for _ in range(4))
        print(name, "  ".join(f"{'auto' if c == 0 else c}: {v:.3f}" for c, v in res.items()), flush=True)
