"""Per-rank kernel time and scheduler census of the bench frame split N ways, on ONE GPU (what each rank of an N-GPU job runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtw_amd as R
scene = R.Scene.generate(R.SCENE_C2, 42)
cam, p = R.default_view(R.SCENE_C5); cam.shutter = 0.0
r = R.Renderer(0); r.set_scene(scene)
out = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
for n in (1, 2, 4, 8):
    for idx in range(n):
        p.row_block, p.part_index, p.part_count = 8, idx, n
        r.render(cam, p, out=out.data_ptr())
        best = None
        for _ in range(3):
            _, st = r.render(cam, p, out=out.data_ptr())
            if best is None or st.kernel_ms < best.kernel_ms: best = st
        eff = [best.phase_lanes[k] / (64.0 * max(1, best.phase_steps[k])) for k in range(3)]
        print(f"N={n} rank={idx}: {best.kernel_ms:.3f} ms  {best.segments / best.kernel_ms / 1e6:.2f} Gseg/s  lanes T/L/S {eff[0]:.3f} {eff[1]:.3f} {eff[2]:.3f}", flush=True)
