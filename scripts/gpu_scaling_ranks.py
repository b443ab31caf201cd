import os, sys
sys.path.insert(0, os.getcwd())
import torch
import rtw_amd as R
scene = R.Scene.generate(R.SCENE_C2, 42)
cam, p = R.default_view(R.SCENE_C5); cam.shutter = 0.0
r = R.Renderer(0); r.set_scene(scene)
out = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
for n in (1, 8):
    for idx in ((0,) if n == 1 else (0, 1, 2, 3, 4, 5, 6, 7)):
        p.row_block, p.part_index, p.part_count = 8, idx, n
        r.render(cam, p, out=out.data_ptr())
        best = min(r.render(cam, p, out=out.data_ptr())[1].kernel_ms for _ in range(3))
        _, st = r.render(cam, p, out=out.data_ptr())
        print(f"chunk={os.environ.get('RTW_CHUNK','4')} N={n} rank={idx}: {best:.3f} ms  segments {st.segments/1e6:.1f}M  -> {st.segments/best/1e6:.2f} Gseg/s", flush=True)
