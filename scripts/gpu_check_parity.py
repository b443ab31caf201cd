"""Quick parity gate for kernel experiments: a few scenes, both path-per-lane settings, against the oracle."""
import sys, numpy as np
sys.path.insert(0, ".")
import rtw_amd as R
from tests import oracle_binding as O
from tests.test_oracle_golden import small_view
ok = True
with R.Renderer(0) as r:
    for which, (w, h, spp) in ((R.SCENE_C2, (160, 90, 16)), (R.SCENE_C4, (96, 54, 8)), (R.SCENE_C5, (96, 54, 8)), (R.SCENE_C1, (64, 36, 4))):
        scene, cam, p = small_view(which, w, h, spp)
        p.gamma = 1.0
        ref, st_ref = O.render(cam, scene, p, 16)
        r.set_scene(scene, cam.time0, cam.time0 + cam.shutter)
        r.set_option(R.OPT_LIST_WALK_MAX, 0)
        for ppl in (1,):
            for flags in (0, R.FLAG_GLOBAL_NODES):
                p.accel, p.flags = R.ACCEL_BVH, flags
                img, st = r.render(cam, p)
                bad = int((np.abs(img - ref).max(axis=2) > 0).sum())
                good = st.segments == st_ref.segments and (bad == 0 or (which == R.SCENE_C5 and bad < 20))
                ok &= good
                print(f"scene {which} paths/lane {ppl} flags {flags}: segments {st.segments} vs {st_ref.segments}, differing pixels {bad} -> {'OK' if good else 'BROKEN'}", flush=True)
print("PARITY", "OK" if ok else "BROKEN")
sys.exit(0 if ok else 1)
