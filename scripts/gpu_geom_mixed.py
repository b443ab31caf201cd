"""The GEOM build of the traversal kernel on a scene that needs it: the Book-1 final scene (485 spheres, BASELINE configs[1] framing: 1200 x 675 x 100 spp, depth 50)
plus a light quad, a mirror quad behind the field, a glass-pane-like quad and a rotated smoke box -- so quads, an instance and a constant-density medium are
tested for every segment while the spheres go through the tree.  For the library selected by RTW_HIP_LIB: kernel time (best of 3) and the md5 of the image."""
import os, sys, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import rtw_amd as R
base = R.Scene.generate(R.SCENE_C2, 42)
spheres = [R.RtwSphere.from_buffer_copy(base._spheres[i]) for i in range(base.n_spheres)]
quads = [R.Quad.new((-2.0, 6.0, -2.0), (4, 0, 0), (0, 0, 4), (0.0, 0.0, 1.0), (1, 1, 1), emitted=(7, 7, 7)),
         R.Quad.new((-8.0, 0.0, -9.0), (16, 0, 0), (0, 5, 0), R.METALLIC_M, (0.8, 0.85, 0.88)),
         R.Quad.new((2.0, 0.0, 2.5), (1.5, 0, -1.0), (0, 1.5, 0), R.GLASS_M, (1, 1, 1))]
box = R.Instance.new_box((-1.0, 0.0, -1.0), (1.0, 1.6, 1.0), (0.9, 0.9, 0.9), R.SCATTER_M)
box.rotate((0.0, 0.5, 0.0)); box.translate((-3.0, 0.0, 3.0)); box.const_density(0.8)
scene = R.Scene(spheres, background=(0.5, 0.7, 1.0), quads=quads, instances=[box])
cam, p = R.default_view(R.SCENE_C2)
r = R.Renderer(0)
r.set_scene(scene, cam.time0, cam.time0 + cam.shutter)
out = torch.zeros((p.height, p.width, 3), dtype=torch.float32, device="cuda:0")
for integ, name in ((R.INTEGRATOR_GRADIENT, "gradient"), (R.INTEGRATOR_BG_COLOR, "bg_color")):
    p.integrator = integ
    r.render(cam, p, out=out.data_ptr())
    best = min((r.render(cam, p, out=out.data_ptr())[1] for _ in range(3)), key=lambda st: st.kernel_ms)
    print(f"{os.environ.get('RTW_HIP_LIB', 'default').split('/')[-1]:24s} book1+geom {name:9s} {best.kernel_ms:9.3f} ms  {best.segments / best.kernel_ms / 1e6:6.2f} G segments/s  "
          f"{best.segments / best.camera_rays:.2f} segments per camera ray  nan pixels {best.nan_pixels}  image md5 {hashlib.md5(out.cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
