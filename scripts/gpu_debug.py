import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtw_amd as R
from tests import oracle_binding as O
r = R.Renderer(0)
for which in (R.SCENE_C1, R.SCENE_C2):
    sc = R.Scene.generate(which); cam, p = R.default_view(which)
    if which == R.SCENE_C2: p.samples = 4
    p.gamma = 1.0
    ref, st = O.render(cam, sc, p, 16)
    r.set_scene(sc)
    print("oracle", st.segments)
    for name, accel, flags in (("brute", 0, 0), ("bvh-lds", 1, 0), ("bvh-global", 1, 4)):
        p.accel, p.flags = accel, flags
        img, s = r.render(cam, p)
        print(name, s.segments, s.node_tests, s.sphere_tests, "equal", np.array_equal(img, ref), "maxdiff", float(np.nanmax(np.abs(img - ref))), flush=True)
