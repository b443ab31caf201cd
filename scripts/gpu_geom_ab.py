import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rtw_amd as R
from tests import oracle_binding as O
out = torch.zeros((400, 400, 3), dtype=torch.float32, device="cuda:0")
ok = True
for name, which in (("quad_test", R.SCENE_QUAD_TEST), ("presentation", R.SCENE_PRESENTATION)):
    sc = R.Scene.generate_geom(which); cam, p = R.default_view(which)
    with R.Renderer(0) as r:
        r.set_scene(sc)
        q = R.RtwParams.from_buffer_copy(p); q.samples = 16; q.gamma = 1.0
        ref, st_ref = O.render(cam, sc, q, 16)
        for accel in (R.ACCEL_BRUTE, R.ACCEL_BVH):
            q.accel = accel; r.set_option(R.OPT_LIST_WALK_MAX, 0)
            img, st = r.render(cam, q)
            same = np.array_equal(np.isnan(img), np.isnan(ref)) and np.array_equal(img[~np.isnan(ref)], ref[~np.isnan(ref)]) and st.segments == st_ref.segments
            ok &= same
        r.set_option(R.OPT_LIST_WALK_MAX, 48)
        best = min(r.render(cam, p, out=out.data_ptr())[1].kernel_ms for _ in range(3))
        _, st = r.render(cam, p, out=out.data_ptr())
        print(f"{name}: {best:.2f} ms {st.segments / best / 1e6:.2f} Gseg/s parity {'OK' if ok else 'BROKEN'}", flush=True)
