import os, sys
sys.path.insert(0, os.getcwd())
import torch
import rtw_amd as R
scene = R.Scene.generate(R.SCENE_C2, 42)
cam, p = R.default_view(R.SCENE_C5); cam.shutter = 0.0
r = R.Renderer(0); r.set_scene(scene)
out = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
rate = None      # G segments/s of the whole frame: what "no fixed cost" would mean for its parts
for n in (1, 8, 16, 32, 64, 135):
    p.row_block, p.part_index, p.part_count = 8, n // 2, n
    r.render(cam, p, out=out.data_ptr())
    best = None
    for _ in range(3):
        _, st = r.render(cam, p, out=out.data_ptr())
        if best is None or st.kernel_ms < best.kernel_ms: best = st
    if rate is None: rate = best.segments / best.kernel_ms / 1e6
    print(f"1/{n} of the rows ({best.rows} rows): {best.kernel_ms:.3f} ms, {best.segments/1e6:.1f} Mseg, at the whole frame's {rate:.2f} G/s {best.segments/rate/1e6:.3f} ms -> overhead {best.kernel_ms - best.segments/rate/1e6:.3f} ms", flush=True)
