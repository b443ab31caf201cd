"""The safety valve of the persistent loops: a library built with -DRTW_MAX_TRIPS=1000 must come back with RTW_E_INTERNAL (-7), not hang."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtw_amd as R
scene = R.Scene.generate(R.SCENE_C2); cam, p = R.default_view(R.SCENE_C2)
with R.Renderer(0) as r:
    r.set_scene(scene)
    for accel in (R.ACCEL_BVH, R.ACCEL_BRUTE):
        p.accel = accel
        try:
            r.render(cam, p); print("accel", accel, "completed (unexpected with 1000 trips)")
        except R.RtwError as e:
            print("accel", accel, "->", e.status, str(e)[:90])
