"""Per-GPU kernel time of the bench frame when the rows are split N ways (what each rank of an N-GPU job runs),
measured on ONE GPU: predicts the strong-scaling efficiency before the 8-GPU node is available."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtw_amd as R
scene = R.Scene.generate(R.SCENE_C2, 42)
cam, p = R.default_view(R.SCENE_C5); cam.shutter = 0.0
r = R.Renderer(0); r.set_scene(scene)
out = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
base = None
for n in (1, 2, 4, 8):
    worst = 0.0
    for idx in (range(n) if n <= 4 else (0, 3, 7)):
        p.row_block, p.part_index, p.part_count = 8, idx, n
        r.render(cam, p, out=out.data_ptr())
        _, st = r.render(cam, p, out=out.data_ptr())
        worst = max(worst, st.kernel_ms)
    base = base or worst
    print(f"N={n}: slowest rank kernel {worst:.2f} ms -> speedup {base / worst:.2f}x, efficiency {base / worst / n:.2%}", flush=True)
