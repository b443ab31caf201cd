"""Exploration: very large sphere counts -- the tree against the list walk (no oracle: the CPU would take minutes)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import rtw_amd as R
r = R.Renderer(0)
vp = R.Viewport.new_from_res(96, 54, 2, 8, 1.0, vfov=70.0, lens_radius=0.01)
cam = vp.camera(); p = vp.params(R.INTEGRATOR_GRADIENT, R.SAMPLER_ROW)
mats = [R.SCATTER_M, R.METALLIC_M, R.GLASS_M]
for n in (20000, 200000, 1000000):
    rng = np.random.default_rng(n)
    c = rng.uniform(-30, 30, (n, 3)).astype(np.float32) + np.float32([0, 0, -40]); rad = rng.uniform(0.02, 0.25, n).astype(np.float32)
    pods = (R.RtwSphere * n)()
    base = R.Sphere.with_albedo((0, 0, 0), 1.0, (0.7, 0.6, 0.5), R.SCATTER_M).pod
    t0 = time.time()
    arr = np.frombuffer(pods, dtype=np.uint8).reshape(n, -1)
    arr[:] = np.frombuffer(bytes(base), dtype=np.uint8)
    f = np.frombuffer(pods, dtype=np.float32).reshape(n, -1)
    f[:, 0:3] = c; f[:, 3] = rad
    scene = R.Scene.__new__(R.Scene)
    try:
        scene = R.Scene(list(pods)) if n <= 20000 else None
    except Exception as e:
        scene = None
    if scene is None:      # large counts: build the RtwScene by hand around the array
        scene = R.Scene([R.Sphere.with_albedo((0, 0, 0), 1.0, (0.7, 0.6, 0.5), R.SCATTER_M)])
        scene._spheres = pods; scene.n_spheres = n
        scene.pod.spheres = pods; scene.pod.n_spheres = n
    t1 = time.time(); r.set_scene(scene); t2 = time.time()
    out = {}
    for label, walk_max, accel in (("tree", 0, R.ACCEL_BVH), ("list", 48, R.ACCEL_BRUTE)):
        r.set_option(R.OPT_LIST_WALK_MAX, walk_max); p.accel = accel
        out[label] = r.render(cam, p)
    (a, sa), (b, sb) = out["tree"], out["list"]
    print(f"n {n}: scene {t1 - t0:.1f} s, set_scene (BVH build + upload) {t2 - t1:.2f} s; tree {sa.kernel_ms:.2f} ms, list {sb.kernel_ms:.1f} ms; images {'==' if np.array_equal(a, b, equal_nan=True) else 'DIFFER'}, segments {sa.segments == sb.segments}, node tests {sa.node_tests}", flush=True)
