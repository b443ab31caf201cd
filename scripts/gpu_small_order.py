import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtw_amd as R
out = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
for name, which in (("quad_test", R.SCENE_QUAD_TEST), ("presentation", R.SCENE_PRESENTATION), ("first_frame", R.SCENE_FIRST_FRAME), ("C1", R.SCENE_C1), ("metal", R.SCENE_METAL_TEST)):
    sc = R.Scene.generate_geom(which) if which in (R.SCENE_QUAD_TEST, R.SCENE_PRESENTATION) else R.Scene.generate(which)
    cam, p = R.default_view(which)
    with R.Renderer(0) as r:
        r.set_scene(sc)
        res = {}
        for order in (0, 1, 2, 3, 0, 1):
            r.set_option(R.OPT_TILE_ORDER, order)
            res.setdefault(order, []).append(min(r.render(cam, p, out=out.data_ptr())[1].kernel_ms for _ in range(5)))
        print(f"{name:14s} " + "  ".join(f"order {o}: {min(v):7.3f}" for o, v in res.items()), flush=True)
