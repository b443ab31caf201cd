"""What costs C5 its 23 % against C3?  The same frame with motion and the image texture switched on one at a time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtw_amd as R
out = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
cam, p = R.default_view(R.SCENE_C5)
with R.Renderer(0) as r:
    for name, moving, textured in (("static, untextured (C3)", 0, 0), ("moving only", 1, 0), ("textured only", 0, 1), ("moving + textured (C5)", 1, 1)):
        sc = R.Scene.generate(R.SCENE_C5)
        for i in range(sc.n_spheres):
            s = sc._spheres[i]
            if not moving:
                s.velocity[0] = s.velocity[1] = s.velocity[2] = 0.0
            if not textured and s.tex >= 0:
                s.tex = -1; s.tex_color[0] = s.tex_color[1] = s.tex_color[2] = 0.5
        c = R.RtwCamera.from_buffer_copy(cam)
        if not moving: c.shutter = 0.0
        r.set_scene(sc, c.time0, c.time0 + c.shutter)
        best = None
        for _ in range(3):
            _, st = r.render(c, p, out=out.data_ptr())
            if best is None or st.kernel_ms < best.kernel_ms: best = st
        eff = [best.phase_lanes[k] / (64.0 * max(1, best.phase_steps[k])) for k in range(3)]
        print(f"{name:28s} {best.kernel_ms:8.2f} ms  {best.segments / best.kernel_ms / 1e6:6.2f} Gseg/s  seg/ray {best.segments / best.camera_rays:.3f}  node visits/seg {best.node_tests / best.segments:.2f} "
              f"leaf+big tests/seg {best.sphere_tests / best.segments:.2f}  eff T/L/S {eff[0]:.3f}/{eff[1]:.3f}/{eff[2]:.3f}  steps {best.phase_steps[0] / 1e6:.0f}M/{best.phase_steps[1] / 1e6:.0f}M/{best.phase_steps[2] / 1e6:.0f}M", flush=True)
