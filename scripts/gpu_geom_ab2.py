"""presentation_image and quad_test (the reference's own quad / instance scenes) for the library selected by RTW_HIP_LIB: kernel time, best of 3, and an
md5 of the image so that variants can be compared bit for bit."""
import os, sys, hashlib
sys.path.insert(0, os.getcwd())
import torch
import rtw_amd as R
r = R.Renderer(0)
for name, which in (("presentation_image", R.SCENE_PRESENTATION), ("quad_test", R.SCENE_QUAD_TEST)):
    sc = R.Scene.generate_geom(which); cam, p = R.default_view(which)
    r.set_scene(sc, cam.time0, cam.time0 + cam.shutter)
    out = torch.zeros((p.height, p.width, 3), dtype=torch.float32, device="cuda:0")
    for walk_max, label in ((8, "as shipped"), (0, "tree forced")):      # RTW_OPT_LIST_WALK_MAX: scenes of <= 8 spheres go to the list walk
        r.set_option(R.OPT_LIST_WALK_MAX, walk_max)
        r.render(cam, p, out=out.data_ptr())
        best = min((r.render(cam, p, out=out.data_ptr())[1] for _ in range(3)), key=lambda st: st.kernel_ms)
        print(f"{os.environ.get('RTW_HIP_LIB', 'default').split('/')[-1]:28s} {name:20s} {label:12s} {best.kernel_ms:9.3f} ms  {best.segments / best.kernel_ms / 1e6:6.2f} G segments/s  image md5 {hashlib.md5(out.cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
