import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import rtw_amd as R
r = R.Renderer(0)
sc = R.Scene.generate_geom(R.SCENE_PRESENTATION); cam, p = R.default_view(R.SCENE_PRESENTATION)
p.samples = 64; p.gamma = 1.0
r.set_scene(sc, cam.time0, cam.time0 + cam.shutter)
imgs = []
for walk_max in (8, 0):
    r.set_option(R.OPT_LIST_WALK_MAX, walk_max)
    out = torch.zeros((p.height, p.width, 3), dtype=torch.float32, device="cuda:0")
    st = r.render(cam, p, out=out.data_ptr())[1]
    imgs.append(out.cpu().numpy()); print(walk_max, st.segments, st.sphere_tests, st.nan_pixels)
a, b = imgs
ab, bb = a.view(np.uint32), b.view(np.uint32)
d = ab != bb
print("differing words", d.sum(), "of which both NaN", (np.isnan(a) & np.isnan(b) & d).sum(), "one NaN", ((np.isnan(a) ^ np.isnan(b)) & d).sum())
for i in np.argwhere(d)[:12]: print(i, a[tuple(i)], b[tuple(i)])
np.save("gpurun_out/geom_modes_list.npy", a); np.save("gpurun_out/geom_modes_tree.npy", b)
