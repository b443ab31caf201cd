"""Quick bit-exactness check of a variant library against the oracle (C2 small, C5 small)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtw_amd as R
from tests import oracle_binding as O
from tests.test_oracle_golden import small_view
r = R.Renderer(0)
ok = True
for which in (R.SCENE_C2, R.SCENE_C5, R.SCENE_C4):
    sc, cam, p = small_view(which, 160, 90, 8); p.gamma = 1.0; p.accel = R.ACCEL_BVH
    ref, st = O.render(cam, sc, p, 16)
    r.set_scene(sc, cam.time0, cam.time0 + cam.shutter)
    img, s = r.render(cam, p)
    good = s.segments == st.segments and (np.abs(img - ref).max(axis=2) > 0).sum() <= (30 if which == R.SCENE_C5 else 0)
    ok &= bool(good)
print("variant", os.environ.get("RTW_HIP_LIB", "default").split("/")[-1], "PARITY", "OK" if ok else "BROKEN")
