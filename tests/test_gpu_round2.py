"""Round-2 additions on the HIP path (through the C ABI): the native multi-GPU entry point, the C++ twin's dialect,
the reference-held fixtures compared DIRECTLY with the GPU output, BVH depth bound, time-range guard, one block of the
headline frame against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

import rtw_amd as R
from tests import oracle_binding as O
from tests.test_oracle_golden import small_view, flag_params, GOLD
from tests.test_abi_and_host import geometric_scene

pytestmark = pytest.mark.gpu


# ---- rtw_mgpu: fork / ordered join of the row tasks over GPUs (Rust/src/viewport.rs:236-244) ------------------------
@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0], [0] * 5])
def test_mgpu_contexts_on_one_gpu_reassemble_the_unsplit_frame(gpu, devices):
    """N contexts (here all on GPU 0) render interleaved 8-row blocks and copy them straight into their image rows: the
    frame is bit-identical to the one-context render; 54 rows = 6 full blocks + a ragged one of 6 rows."""
    scene, cam, p = small_view(R.SCENE_C2, 96, 54, 4)
    gpu.set_scene(scene)
    full, st_full = gpu.render(cam, p)
    with R.MultiRenderer(devices) as m:
        m.set_scene(scene)
        img, tot, per = m.render(cam, p)                                  # host frame
        assert np.array_equal(img, full)
        assert tot.segments == st_full.segments and tot.camera_rays == st_full.camera_rays and tot.rows == 54
        assert sum(s.rows for s in per) == 54 and len(per) == len(devices)
        assert [s.rows for s in per] == [R.lib().rtw_part_rows(54, 8, k, len(devices)) for k in range(len(devices))]
        import torch
        t = torch.full((54, 96, 3), -1.0, dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        m.render(cam, p, out=t.data_ptr())                                # frame in GPU 0's HBM
        assert np.array_equal(t.cpu().numpy(), full)
        q = R.RtwParams.from_buffer_copy(p)                               # other block heights, incl. one that leaves devices idle
        for rb in (1, 16, 64):
            q.row_block = rb
            img2, tot2, _ = m.render(cam, q)
            assert np.array_equal(img2, full) and tot2.segments == st_full.segments, rb
        q.row_block, q.part_index, q.part_count = 8, 0, 2                 # a partition of a partition is refused
        with pytest.raises(R.RtwError):
            m.render(cam, q)


def test_mgpu_one_shot_and_headline_frame(gpu):
    """rtw_render_multi_gpu (create / set scene / render / destroy) on C1, and three contexts on the headline frame
    (1920 x 1080, reduced to 20 spp) against the one-context render."""
    L = R.lib()
    scene = R.Scene.generate(R.SCENE_C1)
    cam, p = R.default_view(R.SCENE_C1)
    gpu.set_scene(scene)
    full, st = gpu.render(cam, p)
    out = np.zeros_like(full)
    per = (R.RtwStats * 3)()
    dev = (C.c_int * 3)(0, 0, 0)
    assert L.rtw_render_multi_gpu(dev, 3, C.byref(cam), C.byref(scene.pod), C.byref(p), out.ctypes.data_as(C.c_void_p), per) == 0
    assert np.array_equal(out, full) and sum(s.segments for s in per) == st.segments
    assert L.rtw_render_multi_gpu(dev, 3, C.byref(cam), C.byref(scene.pod), C.byref(p), None, per) == -1
    bad = (C.c_int * 2)(0, 99)
    assert L.rtw_render_multi_gpu(bad, 2, C.byref(cam), C.byref(scene.pod), C.byref(p), out.ctypes.data_as(C.c_void_p), None) == -2

    scene = R.Scene.generate(R.SCENE_C2)
    cam, p = R.default_view(R.SCENE_C5)
    cam.shutter, p.samples = 0.0, 20
    gpu.set_scene(scene)
    full, st = gpu.render(cam, p)
    with R.MultiRenderer([0, 0, 0]) as m:
        m.set_scene(scene)
        img, tot, _ = m.render(cam, p)
    assert np.array_equal(img, full) and tot.segments == st.segments and tot.camera_rays == 1920 * 1080 * 20


# ---- a11: the C++ twin's dialect on the device ---------------------------------------------------------------------------
@pytest.mark.parametrize("name,mat,flags", [("control", R.METALLIC_M, 0), ("glass", R.GLASS_M, R.FLAG_CPP_DIELECTRIC),
                                            ("glass", R.GLASS_M, R.FLAG_CPP)])
def test_s_test_images_on_the_gpu(gpu, name, mat, flags):
    """The reference's own parity definition (identical 8-bit images of an RNG-free scene, glass_tests.rs:146-193) with the
    HIP path on one side: both 300 x 200 s_test images (C++/src/tests.cpp:275-294) reproduced pixel for pixel."""
    z = np.load(os.path.join(GOLD, "s_test.npz"))
    h, w = [int(x) for x in z[name + "_shape"]]
    want = np.unpackbits(z[name + "_bits"])[: h * w].reshape(h, w).astype(bool)
    scene = R.Scene([R.Sphere.new((0.0, 0.0, -1.0), 0.5, (1.0, 1.0, 1.0), mat),
                     R.Sphere.with_albedo((0.0, -100.5, -1.0), 100.0, (0.8, 0.5, 1.0), R.SCATTER_M)])
    cam, hh = O.viewport_new(300, 1.5)
    assert hh == h
    p = flag_params(flags=flags)
    p.width, p.height = w, h
    gpu.set_scene(scene)
    for accel in (R.ACCEL_BRUTE, R.ACCEL_BVH):
        p.accel = accel
        img, _ = gpu.render(cam, p)
        q = (255.0 * img.astype(np.float64)).astype(np.int32)         # RGB_int: static_cast<int>(255 * c) RGB.cpp:16-20
        yellow = (q == np.array([255, 255, 0])).all(axis=2)
        blue = (q == np.array([0, 0, 255])).all(axis=2)
        assert (yellow | blue).all()
        assert int((yellow != want).sum()) == 0


def test_cerr_trace_pixels_on_the_gpu(gpu):
    """Rust/cerr (the 10 x 10 glass scene of an older C++ build): the pixels whose paths the trace shows ending in the
    sky / on the ground, rendered by the GPU with the deterministic dielectric, agree with the oracle bit for bit and are
    blue / yellow exactly where the trace's last record says sky / scatter."""
    import json
    data = json.load(open(os.path.join(GOLD, "cerr_trace.json")))
    scene = R.Scene([R.Sphere.new((0.0, 0.0, -1.0), 0.5, (1.0, 1.0, 1.0), R.GLASS_M),
                     R.Sphere.new((0.0, -100.5, -1.0), 100.0, (1.0, 1.0, 1.0), R.EMPTY_M)])
    # the old camera: dir = (-1 + 2u, -1 + 2v, -1), u = x/9, v = (9-y)/9  ==  pixel00 + x du + y dv
    cam = R.RtwCamera()
    f = np.float32
    for k, v in enumerate((-1.0, 1.0, -1.0)):
        cam.pixel00[k] = v
    cam.delta_u[0] = float(f(2) / f(9))
    cam.delta_v[1] = -float(f(2) / f(9))
    p = flag_params(flags=R.FLAG_CPP_DIELECTRIC)
    p.gamma = 1.0
    ref, _ = O.render(cam, scene, p, threads=1)
    gpu.set_scene(scene)
    img, _ = gpu.render(cam, p)
    assert np.array_equal(img, ref)
    checked = 0
    for px in data["pixels"]:
        last = px["bounces"][-1]["kind"]
        got = tuple(img[px["y"], px["x"]])
        assert got == ((0.0, 0.0, 1.0) if last == "sky" else (1.0, 1.0, 0.0)), px
        checked += 1
    assert checked == 58


def test_cpp_dialect_bit_exact_and_statistical(gpu):
    """RTW_FLAG_CPP_DIELECTRIC | RTW_FLAG_CPP_DIFFUSE on the HIP path == the oracle bit for bit (both closest-hit
    strategies), and -- where oracle/_ref exists -- statistically == the reference's own C++ objects."""
    glass = R.Sphere.with_albedo((-1.0, 0.0, -1.0), 0.5, (1, 1, 1), R.GLASS_M)
    scene = R.Scene.generate(R.SCENE_METAL_TEST)
    spheres = [scene._spheres[i] for i in range(scene.n_spheres)] + [glass.pod]
    scene = R.Scene(spheres)
    cam, h = O.viewport_new(96, np.float32(96) / np.float32(54))
    cam.lens_radius = 0.01
    p = flag_params(depth=10, flags=R.FLAG_CPP)
    p.width, p.height, p.samples, p.integrator, p.sampler, p.gamma = 96, 54, 192, R.INTEGRATOR_GRADIENT, R.SAMPLER_ROW, 1.0
    ref, st_ref = O.render(cam, scene, p, threads=16)
    gpu.set_scene(scene)
    imgs = {}
    for accel in (R.ACCEL_BRUTE, R.ACCEL_BVH):
        p.accel = accel
        img, st = gpu.render(cam, p)
        assert st.segments == st_ref.segments
        assert np.array_equal(img, ref), accel
        imgs[accel] = img
    p.flags = 0
    plain, _ = gpu.render(cam, p)
    assert not np.array_equal(plain, ref)
    if O.have_ref():
        p.flags = R.FLAG_CPP
        b, seg, _ = O.ref_render(cam, scene, p, rand_seed=5)

        def blocks(x):
            return x[:54, :96].reshape(9, 6, 16, 6, 3).mean(axis=(1, 3))
        da = blocks(imgs[R.ACCEL_BVH].astype(np.float64)) - blocks(b)
        assert abs(da.mean()) < 2e-3 and np.abs(da).max() < 0.05, (da.mean(), np.abs(da).max())


def test_rust2_integrator_honours_cpp_dielectric(gpu):
    from tests.test_oracle_golden import rust2_view
    scene, cam, p = rust2_view(samples=16)
    p.flags, p.gamma = R.FLAG_CPP_DIELECTRIC, 1.0
    ref, st_ref = O.render(cam, scene, p)
    gpu.set_scene(scene)
    img, st = gpu.render(cam, p)
    assert st.segments == st_ref.segments and np.array_equal(img, ref)


# ---- Vec3::rotated known answers (vec3.rs:363-404) through the GPU ---------------------------------------------------------
@pytest.mark.parametrize("rot,want", [((np.pi / 6, 0.0, 0.0), (1.0, 0.0, -0.0)), ((0.0, 0.0, np.pi / 6), None)])
def test_rotated_known_answers_on_the_gpu(gpu, rot, want):
    """An instance holding a quad with normal (1, 0, 0), rotated by the reference's test rotations; the NORMAL integrator
    returns (rotated(normal) + 1) / 2, so the image exposes Vec3::rotated as the device applies it."""
    rot = tuple(float(np.float32(x)) for x in rot)
    inst = R.Instance.new_quads([R.Quad.new((3, -50, -50), (0, 100, 0), (0, 0, 100), R.SCATTER_M, (1, 1, 1))])
    inst.rotate(rot)
    scene = R.Scene.new([], [], [inst])
    cam, h = O.viewport_new(32, np.float32(32) / np.float32(18), origin=(0, 0, 0), direction=(1, 0, 0))
    p = flag_params(depth=2)
    p.width, p.height, p.samples, p.integrator, p.sampler, p.gamma = 32, 18, 1, R.INTEGRATOR_NORMAL, R.SAMPLER_NO_RAND, 1.0
    ref, _ = O.render(cam, scene, p)
    gpu.set_scene(scene)
    for accel in (R.ACCEL_BRUTE, R.ACCEL_BVH):
        p.accel = accel
        img, st = gpu.render(cam, p)
        assert np.array_equal(img, ref)
    n = img[9, 16].astype(np.float64) * 2.0 - 1.0
    if want is None:
        want = (np.cos(np.float32(rot[2])), np.sin(np.float32(rot[2])), 0.0)          # vec3.rs:385-391
    assert np.abs(n - np.array(want, np.float64)).max() < 2e-7, (n, want)
    assert np.abs(n - R.vec3_rotated((1, 0, 0), rot)).max() < 1.2e-7


# ---- ADVICE r1 ---------------------------------------------------------------------------------------------------------------
def test_skewed_scene_stays_inside_the_device_stack(gpu):
    """Spheres at x = 1.2^i (the old "SAH until depth 20" rule built depth 27 and overflowed the 24-level LDS
    stack): the tree now stops at depth 24 exactly, and both node variants return the list walk's image."""
    sc = geometric_scene(200, 1.2, rel_radius=0.12)
    depth = C.c_uint32()
    assert R.lib().rtw_bvh_validate(C.byref(sc.pod), 0.0, 0.0, None, C.byref(depth), None, None) == 0 and depth.value == 24
    cam, h = O.viewport_new(96, np.float32(96) / np.float32(54), origin=(6.0, 0.3, 2.0), direction=(0.3, -0.03, -1.0), vfov=80.0)
    p = flag_params(depth=6)
    p.width, p.height, p.samples, p.integrator, p.sampler, p.gamma = 96, 54, 8, R.INTEGRATOR_GRADIENT, R.SAMPLER_ROW, 1.0
    ref, st_ref = O.render(cam, sc, p, threads=16)
    assert st_ref.segments > 1.1 * st_ref.camera_rays
    gpu.set_scene(sc)
    for flags in (0, R.FLAG_GLOBAL_NODES):
        p.accel, p.flags = R.ACCEL_BVH, flags
        img, st = gpu.render(cam, p)
        assert st.segments == st_ref.segments and np.array_equal(img, ref), flags
        assert st.node_tests > 0


def test_moving_scene_outside_the_built_time_range(gpu):
    """The BVH bounds cover the ray.time range given to rtw_ctx_set_scene; a render outside it must not prune with them."""
    scene, cam, p = small_view(R.SCENE_C5, 96, 54, 8)
    p.gamma = 1.0
    cam.time0, cam.shutter = 2.0, 0.5                      # the moving spheres have long left their t in [0, 1/30] boxes
    ref, st_ref = O.render(cam, scene, p, threads=16)
    gpu.set_scene(scene, 0.0, 1.0 / 30.0)                  # built for frame 0
    p.accel = R.ACCEL_BVH
    img, st = gpu.render(cam, p)
    assert st.segments == st_ref.segments and st.node_tests == 0          # demoted to the list walk
    assert (np.abs(img - ref).max(axis=2) > 0).sum() <= 0.002 * 96 * 54       # (textured ground: atan2f / acosf texel edges)
    gpu.set_scene(scene, 2.0, 2.5)
    img2, st2 = gpu.render(cam, p)
    assert st2.node_tests > 0 and np.array_equal(img2, img)


def test_small_scenes_walk_the_list(rtw):
    """RTW_OPT_LIST_WALK_MAX: a BVH request on a handful of spheres runs the list-walk kernel (same image)."""
    scene = R.Scene.generate(R.SCENE_C1)
    cam, p = R.default_view(R.SCENE_C1)
    p.accel = R.ACCEL_BVH
    with rtw.Renderer(0) as r:
        r.set_scene(scene)
        a, sa = r.render(cam, p)
        assert sa.node_tests == 0                                        # default: 3 spheres are below the crossover
        r.set_option(R.OPT_LIST_WALK_MAX, 0)
        b, sb = r.render(cam, p)
        assert sb.node_tests > 0 and np.array_equal(a, b) and sa.segments == sb.segments


# ---- the headline frame against the oracle -------------------------------------------------------------------------------------
def test_c3_one_block_against_the_oracle(gpu):
    """BASELINE config 3 (1920 x 1080 x 500 spp, depth 50): one 8-row block through the sphere field (rows 600..607,
    7.7 M camera rays) against the oracle -- bit for bit at gamma 1, <= 2 ulp (powf) at the reference's gamma 2."""
    scene = R.Scene.generate(R.SCENE_C2)
    cam, p = R.default_view(R.SCENE_C5)
    cam.shutter = 0.0
    p.row_block, p.part_index, p.part_count = 8, 75, 135            # block 75 = rows 600..607
    p.gamma = 1.0
    ref, st_ref = O.render(cam, scene, p, threads=16)
    gpu.set_scene(scene)
    img, st = gpu.render(cam, p)
    assert st.rows == 8 and st.camera_rays == 8 * 1920 * 500 and st.segments == st_ref.segments
    assert np.array_equal(img, ref)


# ---- RTW_FLAG_CHUNK_SUMS: one partial sum per (pixel, 4 samples) in the bank instead of every sample ---------------------------------
@pytest.mark.parametrize("samples", [1, 3, 4, 10, 37])
def test_chunk_sums_mode(gpu, samples):
    """The chunked association ((s0+s1+s2+s3) + (s4+..)) + .. is implemented by the oracle under the same flag: bit-identical;
    against the reference's left-to-right order it differs by f32 rounding only; the image stays independent of the row split
    and of the work-unit option."""
    scene, cam, p = small_view(R.SCENE_C2, 96, 54, samples)
    p.gamma = 1.0
    ordered, _ = O.render(cam, scene, p, threads=16)
    p.flags = R.FLAG_CHUNK_SUMS
    ref, st_ref = O.render(cam, scene, p, threads=16)
    with R.Renderer(0) as r:
        r.set_scene(scene)
        for accel in (R.ACCEL_BRUTE, R.ACCEL_BVH):
            p.accel = accel
            img, st = r.render(cam, p)
            assert st.segments == st_ref.segments and np.array_equal(img, ref), accel
        r.set_option(R.OPT_CHUNK_LEN, 7)                       # ignored in this mode: the summation chunk defines the image
        img2, _ = r.render(cam, p)
        assert np.array_equal(img2, ref)
        p.row_block, p.part_index, p.part_count = 8, 1, 3
        part, _ = r.render(cam, p)
        rows = [j for j in range(54) if (j // 8) % 3 == 1]
        assert np.array_equal(part, ref[rows])
    rel = np.abs(ref - ordered) / np.maximum(np.abs(ordered), 1e-6)
    assert rel.max() < 2e-6 * max(1, samples // 4), rel.max()
    if samples <= 4:
        assert np.array_equal(ref, ordered)                    # a single chunk IS the left-to-right sum


def test_mgpu_moving_textured_and_geom_scenes(gpu):
    """The native multi-GPU path on everything a scene can hold: motion blur + image texture (C5), and quads / instances /
    smoke with the emissive integrator (presentation_image) -- three contexts reassemble the one-context frame bit for bit."""
    for which, geom in ((R.SCENE_C5, False), (R.SCENE_PRESENTATION, True)):
        if geom:
            scene = R.Scene.generate_geom(which)
            cam, p = R.default_view(which)
            p.samples = 16
        else:
            scene, cam, p = small_view(which, 160, 90, 8)
        gpu.set_scene(scene, cam.time0, cam.time0 + cam.shutter)
        full, st = gpu.render(cam, p)
        with R.MultiRenderer([0, 0, 0]) as m:
            m.set_scene(scene, cam.time0, cam.time0 + cam.shutter)
            img, tot, per = m.render(cam, p)
        assert np.array_equal(np.isnan(img), np.isnan(full))
        ok = ~np.isnan(full)
        assert np.array_equal(img[ok], full[ok]) and tot.segments == st.segments and tot.nan_pixels == st.nan_pixels, which


def test_example_program_multi_gpu_entry(gpu, tmp_path):
    """examples/render_scene --devices 0,0,0: rtw_render_multi_gpu from compiled code, host frame; same PNG as one device."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "render_scene")
    subprocess.run(["make", "-C", os.path.join(root, "raytracing-in-a-weekend_amd", "csrc"), "example"], check=True, capture_output=True)
    a, b = str(tmp_path / "one.png"), str(tmp_path / "three.png")
    subprocess.run([exe, "--scene", "book1", "--width", "320", "--height", "180", "--spp", "8", "--out", a], check=True, capture_output=True)
    out = subprocess.run([exe, "--scene", "book1", "--width", "320", "--height", "180", "--spp", "8", "--devices", "0,0,0", "--out", b],
                         check=True, capture_output=True, text=True).stdout
    assert out.startswith("3 devices: 485 spheres") and "460800 camera rays" in out
    assert open(a, "rb").read() == open(b, "rb").read()


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_tile_order_never_changes_the_image(rtw, mode):
    """RTW_OPT_TILE_ORDER only permutes the work queue: frames, row partitions (whose tiles map to other image rows) and banded
    renders are bit-identical under every order."""
    scene, cam, p = small_view(R.SCENE_C2, 200, 120, 6)
    p.gamma = 1.0
    ref, st_ref = O.render(cam, scene, p, threads=16)
    with rtw.Renderer(0) as r:
        r.set_scene(scene)
        r.set_option(R.OPT_TILE_ORDER, mode)
        for accel in (R.ACCEL_BVH, R.ACCEL_BRUTE):
            p.accel = accel
            img, st = r.render(cam, p)
            assert st.segments == st_ref.segments and np.array_equal(img, ref), (mode, accel)
        p.accel = R.ACCEL_BVH
        r.set_option(R.OPT_SAMPLE_BANK_GB, 0.004)                 # several bands of tile rows
        img, _ = r.render(cam, p)
        assert np.array_equal(img, ref)
        r.set_option(R.OPT_SAMPLE_BANK_GB, 48)
        p.row_block, p.part_index, p.part_count = 8, 2, 3
        part, _ = r.render(cam, p)
        rows = [j for j in range(120) if (j // 8) % 3 == 2]
        assert np.array_equal(part, ref[rows])


def test_bench_single_process_multi_gpu_path(gpu):
    """bench.py --devices 0,0,0: the single-process N-GPU path of the bench (rtw_mgpu, frame assembled in GPU 0's HBM by strided
    copies, no torch.distributed) end to end, with three contexts on this one GPU."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--devices", "0,0,0", "--steps", "1", "--warmup", "1", "--spp", "8", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 3 and d["value"] > 0 and "rtw_mgpu" in d["config"]["workload"]
    # three contexts traced the whole frame between them: 1920 x 1080 x 8 camera rays
    assert abs(d["roofline"]["units_per_launch"]["segments"] * 3 / (1920 * 1080 * 8) - d["config"]["segments_per_camera_ray"]) < 1e-3


def test_persistent_loop_safety_valve(tmp_path):
    """A wave of a persistent kernel that exceeds RTW_MAX_TRIPS scheduler trips gives up and the render returns RTW_E_INTERNAL (-7)
    instead of hanging the GPU: a variant build with a limit of 1000 trips (the C2 frame needs far more) must come back with -7."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = str(tmp_path / "librtw_valve.so")
    subprocess.run(["make", "-s", "-C", os.path.join(root, "raytracing-in-a-weekend_amd", "csrc"), "OUT=" + lib, "EXTRA=-DRTW_MAX_TRIPS=1000u"], check=True, capture_output=True, timeout=600)
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "gpu_valve_check.py")], env=dict(os.environ, RTW_HIP_LIB=lib), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-1500:]
    assert "accel 1 -> -7" in out.stdout, out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("log2_scale", [-31, -14, 12])
def test_sphere_tests_outside_the_plain_range(gpu, log2_scale):
    """sphere_root() takes the plain sqrt / division sequences only while d.d is in [2^-20, 2^20] (and disc in [2^-60, 2^96]); a scene
    scaled by 2^-14 or 2^12 (camera rays with d.d ~ 2^-28 / 2^24) sends every wave down the generic expansions instead, and at 2^-31 the
    scattered rays (d.d ~ 1) meet discriminants around 2^-62 as well, below the per-test range.  All must give
    the oracle's bits, in the tree kernel (sticky per-wave flag) and in the list walk (one ballot per query)."""
    s = float(2.0 ** log2_scale)
    scene, cam, p = small_view(R.SCENE_C2, 96, 54, 4)
    p.gamma, p.depth = 1.0, 8
    for i in range(scene.n_spheres):
        sp = scene._spheres[i]
        for k in range(3): sp.center[k] *= s
        sp.radius *= s
    for name in ("origin", "pixel00", "delta_u", "delta_v"):
        v = getattr(cam, name)
        for k in range(3): v[k] *= s
    cam.lens_radius *= s
    # (the ray parameter t = (-b -+ sqrt(disc)) / a does not change with the scale -- b, sqrt(disc) and a all carry 2^(2 log2_scale) -- so
    # mint / maxt stay as they are)
    ref, st_ref = O.render(cam, scene, p, threads=16)
    assert st_ref.segments > 1.5 * st_ref.camera_rays      # the scaled scene still scatters
    gpu.set_scene(scene)
    for accel in (R.ACCEL_BVH, R.ACCEL_BRUTE):
        p.accel = accel
        img, st = gpu.render(cam, p)
        assert st.segments == st_ref.segments and np.array_equal(img, ref), (log2_scale, accel)
        assert (st.node_tests > 0) == (accel == R.ACCEL_BVH)


@pytest.mark.gpu
@pytest.mark.parametrize("blocks", [0, 1, 2, 5])
def test_grab_size_never_changes_the_image(rtw, blocks):
    """RTW_OPT_GRAB_BLOCKS only changes how many 64-unit blocks a wave takes from the work queue per atomic (0: up to a tile's worth):
    full frames, banded renders and row partitions are bit-identical, with both kernels, for a frame large enough that the guided rule
    really hands out several blocks at a time, and every unit is rendered exactly once (camera-ray count)."""
    scene, cam, p = small_view(R.SCENE_C2, 320, 184, 40)
    p.gamma, p.depth = 1.0, 6
    with rtw.Renderer(0) as r:
        r.set_scene(scene)
        r.set_option(R.OPT_GRAB_BLOCKS, 1)
        p.accel = R.ACCEL_BVH
        ref, st_ref = r.render(cam, p)                            # (single blocks: the behaviour every other parity test pins to the oracle)
        r.set_option(R.OPT_GRAB_BLOCKS, blocks)
        for accel in (R.ACCEL_BVH, R.ACCEL_BRUTE):
            p.accel = accel
            img, st = r.render(cam, p)
            assert st.camera_rays == 320 * 184 * 40 and st.segments == st_ref.segments and np.array_equal(img, ref), (blocks, accel)
        p.accel = R.ACCEL_BVH
        r.set_option(R.OPT_SAMPLE_BANK_GB, 0.01)                  # several bands of tile rows
        img, _ = r.render(cam, p)
        assert np.array_equal(img, ref)
        r.set_option(R.OPT_SAMPLE_BANK_GB, 48)
        p.row_block, p.part_index, p.part_count = 8, 1, 3
        part, _ = r.render(cam, p)
        rows = [j for j in range(184) if (j // 8) % 3 == 1]
        assert np.array_equal(part, ref[rows])


@pytest.mark.gpu
@pytest.mark.parametrize("moving", [False, True])
@pytest.mark.parametrize("textured", [False, True])
def test_every_build_of_the_traversal_kernel(gpu, moving, textured):
    """render_bvh is compiled 28 times (MOVING x NODES {global f32 nodes, LDS f16 nodes, LDS nodes + LDS sphere geometry} x SPEC
    {generic, common, textured, chunk sums} + the GEOM builds); which build a request runs depends on the scene's size, motion and
    textures, on the integrator and on three knobs.  A scene of 130 spheres (small enough for its geometry to fit in LDS, large
    enough to be given to the tree) is rendered through every build it can reach and each image compared with the oracle."""
    rng = np.random.default_rng(5 + 2 * moving + textured)
    tex = rng.uniform(0.1, 0.9, size=(4, 6, 3)).astype(np.float32)
    mats = [R.SCATTER_M, R.METALLIC_M, R.GLASS_M, R.FUZZY3_M]
    ground = R.Sphere.new_with_texture((0, -1000, 0), 1000.0, None, R.SCATTER_M, 0) if textured else R.Sphere.with_albedo((0, -1000, 0), 1000.0, (0.5, 0.5, 0.5))
    spheres = [ground]
    for i in range(129):
        c = (float(rng.uniform(-5, 5)), float(rng.uniform(0.15, 1.0)), float(rng.uniform(-6, 1)))
        vel = (0.0, float(rng.uniform(0.0, 6.0)), 0.0) if (moving and i % 3 == 0) else None
        spheres.append(R.Sphere.with_albedo(c, float(rng.uniform(0.1, 0.3)), tuple(rng.uniform(0.3, 0.9, 3)), mats[i % 4], velocity=vel))
    scene = R.Scene(spheres, textures=[tex] if textured else [])
    vp = R.Viewport.new_from_res(112, 64, 6, 8, 1.0, vfov=45.0, origin=(0.0, 1.6, 6.0), direction=(0.0, -0.2, -1.0), lens_radius=0.03)
    if moving: vp.shutter_speed, vp.fps = 1.0 / 30.0, 30.0
    cam = vp.camera()
    gpu.set_scene(scene, cam.time0, cam.time0 + cam.shutter)

    def close(img, ref):
        if not textured: return np.array_equal(img, ref)
        same = (img == ref).all(axis=2)                       # (atan2f / acosf: a hit may fall on the other side of a texel edge)
        return same.mean() > 0.998 and np.abs(img - ref).max() < 1.0

    seen = set()
    for integrator in (R.INTEGRATOR_GRADIENT, R.INTEGRATOR_NORMAL):          # SPEC 1 / 2 builds, and the generic build
        for flags in (0, R.FLAG_CHUNK_SUMS):                                   # ... SPEC 3 (gradient, untextured) or generic
            p = vp.params(integrator, R.SAMPLER_ROW)
            p.gamma, p.flags = 1.0, flags
            ref, st_ref = O.render(cam, scene, p, threads=16)
            for lds_geom in (0, 1):                                            # NODES 1 / 2
                for extra in (0, R.FLAG_GLOBAL_NODES):                         # NODES 0
                    gpu.set_option(R.OPT_LDS_GEOM, lds_geom)
                    p.accel, p.flags = R.ACCEL_BVH, flags | extra
                    img, st = gpu.render(cam, p)
                    assert st.node_tests > 0 and st.segments == st_ref.segments, (integrator, flags, lds_geom, extra)
                    assert close(img, ref), (integrator, flags, lds_geom, extra)
                    seen.add((integrator, flags, lds_geom, extra))
            p.accel, p.flags = R.ACCEL_BRUTE, flags
            img, st = gpu.render(cam, p)
            assert st.segments == st_ref.segments and close(img, ref)
    gpu.set_option(R.OPT_LDS_GEOM, -1)
    assert len(seen) == 16
