"""Which call of librtw_hip.so makes a LATER torch.cuda initialisation fail ("No HIP GPUs are available")?
Each case runs in its own process."""
import subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {
    "torch_only": "import torch; print(torch.zeros(3, device='cuda:0').sum().item())",
    "ctx_then_torch": "import rtw_amd as R; r = R.Renderer(0); import torch; print(torch.zeros(3, device='cuda:0').sum().item())",
    "ctx_render_then_torch": "import rtw_amd as R; r = R.Renderer(0); s = R.Scene.generate(R.SCENE_C1); cam, p = R.default_view(R.SCENE_C1); r.set_scene(s); r.render(cam, p); import torch; print(torch.zeros(3, device='cuda:0').sum().item())",
    "two_ctx_then_torch": "import rtw_amd as R; r = R.Renderer(0); r2 = R.Renderer(0); import torch; print(torch.zeros(3, device='cuda:0').sum().item())",
    "mgpu_then_torch": "import rtw_amd as R; r = R.MultiRenderer([0]); import torch; print(torch.zeros(3, device='cuda:0').sum().item())",
    "count_then_torch": "import rtw_amd as R; print(R.device_count()); import torch; print(torch.zeros(3, device='cuda:0').sum().item())",
    "torch_imported_first_ctx_then_cuda": "import torch; import rtw_amd as R; r = R.Renderer(0); print(torch.zeros(3, device='cuda:0').sum().item())",
}
for name, code in CASES.items():
    pr = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); " % ROOT + code], capture_output=True, text=True, timeout=300)
    last = (pr.stderr.strip().splitlines() or [""])[-1]
    print(f"{name:40s} rc={pr.returncode} out={pr.stdout.strip()!r} err={last[:150]!r}", flush=True)
