import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtw_amd as R
out = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
for name, sid, vid in (("C3", R.SCENE_C2, R.SCENE_C5), ("C4", R.SCENE_C4, R.SCENE_C4)):
    scene = R.Scene.generate(sid); cam, p = R.default_view(vid)
    if name == "C3": cam.shutter = 0.0
    with R.Renderer(0) as r:
        r.set_scene(scene)
        res = {}
        for chunk in (8, 10, 12, 16, 20, 25, 32):
            r.set_option(R.OPT_CHUNK_LEN, chunk)
            res[chunk] = min(r.render(cam, p, out=out.data_ptr())[1].kernel_ms for _ in range(4))
        print(name, "  ".join(f"{c}: {v:.3f}" for c, v in res.items()), flush=True)
