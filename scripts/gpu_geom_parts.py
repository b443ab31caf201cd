"""What the GEOM build of the traversal kernel pays for, feature by feature: the Book-1 final scene (485 spheres, 1200 x 675 x 100 spp, depth 50) with
nothing / quads / a box instance / the box as a constant-density medium added.  Kernel time (best of 3) for the library selected by RTW_HIP_LIB."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import rtw_amd as R
base = R.Scene.generate(R.SCENE_C2, 42)
spheres = [R.RtwSphere.from_buffer_copy(base._spheres[i]) for i in range(base.n_spheres)]
def quads():
    return [R.Quad.new((-2.0, 6.0, -2.0), (4, 0, 0), (0, 0, 4), (0.0, 0.0, 1.0), (1, 1, 1), emitted=(7, 7, 7)),
            R.Quad.new((-8.0, 0.0, -9.0), (16, 0, 0), (0, 5, 0), R.METALLIC_M, (0.8, 0.85, 0.88)),
            R.Quad.new((2.0, 0.0, 2.5), (1.5, 0, -1.0), (0, 1.5, 0), R.GLASS_M, (1, 1, 1))]
def box(medium):
    b = R.Instance.new_box((-1.0, 0.0, -1.0), (1.0, 1.6, 1.0), (0.9, 0.9, 0.9), R.SCATTER_M)
    b.rotate((0.0, 0.5, 0.0)); b.translate((-3.0, 0.0, 3.0))
    if medium: b.const_density(0.8)
    return b
cam, p = R.default_view(R.SCENE_C2)
r = R.Renderer(0)
out = torch.zeros((p.height, p.width, 3), dtype=torch.float32, device="cuda:0")
cases = (("spheres only, gradient (specialised build)", R.INTEGRATOR_GRADIENT, [], []),
         ("spheres only, bg_color (the generic step, switches folded in)", R.INTEGRATOR_BG_COLOR, [], []),
         ("+ 3 quads", R.INTEGRATOR_GRADIENT, quads(), []),
         ("+ 1 quad", R.INTEGRATOR_GRADIENT, quads()[:1], []),
         ("+ a box instance (6 quads)", R.INTEGRATOR_GRADIENT, [], [box(False)]),
         ("+ the box as a medium", R.INTEGRATOR_GRADIENT, [], [box(True)]),
         ("+ 3 quads + the medium", R.INTEGRATOR_GRADIENT, quads(), [box(True)]))
import numpy as np
cam2 = R.camera2_new(np.float32(p.width) / np.float32(p.height), (13, 2, 3), (0, 1, 0), (-13 / 13.49, -2 / 13.49, -3 / 13.49), 20.0, 0.05)   # Rust2's Camera on the same view
cases += (("spheres only, Rust2's ray_color + render_row", R.INTEGRATOR_RUST2, [], []),)
cam1, sampler1 = cam, p.sampler
for name, integ, q, inst in cases:
    scene = R.Scene(spheres, background=(0.5, 0.7, 1.0), quads=q, instances=inst)
    cam, p.sampler = (cam2, R.SAMPLER_CENTRES) if integ == R.INTEGRATOR_RUST2 else (cam1, sampler1)
    r.set_scene(scene, cam.time0, cam.time0 + cam.shutter)
    p.integrator = integ
    r.render(cam, p, out=out.data_ptr())
    best = min((r.render(cam, p, out=out.data_ptr())[1] for _ in range(3)), key=lambda st: st.kernel_ms)
    print(f"{name:46s} {best.kernel_ms:8.3f} ms  {best.segments / best.kernel_ms / 1e6:6.2f} G segments/s  {best.segments / best.camera_rays:.2f} segments per camera ray  "
          f"{best.quad_tests / best.segments:5.2f} quad tests per segment", flush=True)
