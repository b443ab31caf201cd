"""First GPU contact: parity of both closest-hit strategies vs the oracle, then rough timings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtw_amd as R
from tests import oracle_binding as O

def cmp(name, img, ref):
    eq = np.array_equal(img, ref)
    d = np.abs(img.astype(np.float64) - ref)
    print(f"  {name}: bit-exact={eq} max|d|={d.max():.3e} mismatched px={(d.max(axis=2) > 0).sum()} / {d.shape[0]*d.shape[1]}", flush=True)

print("devices", R.device_count(), flush=True)
r = R.Renderer(0)
for which, name in ((R.SCENE_C1, "C1"), (R.SCENE_METAL_TEST, "metal"), (R.SCENE_C2, "C2-small"), (R.SCENE_C4, "C4-small"), (R.SCENE_C5, "C5-small")):
    sc = R.Scene.generate(which)
    cam, p = R.default_view(which)
    if which in (R.SCENE_C2, R.SCENE_C4, R.SCENE_C5):
        # shrink: same camera basis, fewer pixels -> scale the deltas
        w, h = 240, 135
        f = p.width / w
        for k in range(3):
            cam.pixel00[k] = cam.pixel00[k] - 0.5 * (cam.delta_u[k] + cam.delta_v[k]) + 0.5 * f * (cam.delta_u[k] + cam.delta_v[k])
            cam.delta_u[k] *= f; cam.delta_v[k] *= f
        p.width, p.height, p.samples = w, h, 8
    p.gamma = 1.0
    t = time.time(); ref, st_ref = O.render(cam, sc, p, 16); t_cpu = time.time() - t
    print(f"{name}: n={sc.n_spheres} oracle {t_cpu:.2f}s segs={st_ref.segments}", flush=True)
    r.set_scene(sc, cam.time0, cam.time0 + cam.shutter)
    for accel in (R.ACCEL_BRUTE, R.ACCEL_BVH):
        p.accel = accel
        img, st = r.render(cam, p)
        print(f"  accel={accel} kernel {st.kernel_ms:.2f} ms segs={st.segments} tests={st.sphere_tests} nodes={st.node_tests} nan={st.nan_pixels}", flush=True)
        cmp("vs oracle", img, ref)

# timing at size
for which, name in ((R.SCENE_C2, "C2 full 1200x675x100"),):
    sc = R.Scene.generate(which)
    cam, p = R.default_view(which)
    r.set_scene(sc, cam.time0, cam.time0 + cam.shutter)
    imgs = {}
    for accel in (R.ACCEL_BVH, R.ACCEL_BRUTE):
        p.accel = accel
        img, st = r.render(cam, p)
        imgs[accel] = img
        print(f"{name} accel={accel}: kernel {st.kernel_ms:.1f} ms, {st.segments/st.kernel_ms/1e3:.1f} Mseg/s, {st.camera_rays/st.kernel_ms/1e3:.1f} Mcam/s, "
              f"tests/seg={st.sphere_tests/st.segments:.1f} nodes/seg={st.node_tests/st.segments:.1f}", flush=True)
    cmp("BVH vs brute (GPU)", imgs[R.ACCEL_BVH], imgs[R.ACCEL_BRUTE])
