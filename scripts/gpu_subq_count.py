import os, sys
sys.path.insert(0, os.getcwd())
import torch
import rtw_amd as R
out = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
cases = (("C1", R.SCENE_C1, R.SCENE_C1, 1, None), ("C2", R.SCENE_C2, R.SCENE_C2, 1, None), ("C3 full", R.SCENE_C2, R.SCENE_C5, 1, 0.0), ("C3 1/8", R.SCENE_C2, R.SCENE_C5, 8, 0.0), ("First frame", R.SCENE_FIRST_FRAME, R.SCENE_FIRST_FRAME, 1, None))
res=[]
for name, sid, vid, parts, shutter in cases:
    scene = R.Scene.generate(sid); cam, p = R.default_view(vid)
    if shutter is not None: cam.shutter = shutter
    if parts > 1: p.row_block, p.part_index, p.part_count = 8, 3, parts
    with R.Renderer(0) as r:
        r.set_scene(scene, cam.time0, cam.time0 + cam.shutter)
        best = min(r.render(cam, p, out=out.data_ptr())[1].kernel_ms for _ in range(5))
        res.append(f"{name}: {best:.3f}")
print(os.environ.get("RTW_HIP_LIB","default").split("/")[-1], "  ".join(res), flush=True)
