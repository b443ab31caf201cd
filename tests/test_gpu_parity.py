"""Parity of the HIP path (through the C ABI of librtw_hip.so) against the CPU oracle.

Bar (DESIGN.md "Parity"): with gamma == 1 the whole path is IEEE +,-,*,/,sqrt in the reference's
operation order on both sides, so the images must be BIT-IDENTICAL and the segment counts equal; with
gamma != 1 the only difference allowed is libm powf (glibc vs ocml): <= 2 ulp per channel.
Textured spheres add atan2f/acosf (texel choice): a handful of pixels may pick a neighbouring texel.
north_star's tolerance is per-channel |delta| < 1e-3; what is asserted here is far tighter.
"""
import ctypes as C
import os

import numpy as np
import pytest

import rtw_amd as R
from tests import oracle_binding as O
from tests.test_oracle_golden import small_view, flag_params

pytestmark = pytest.mark.gpu

BOTH = (R.ACCEL_BRUTE, R.ACCEL_BVH)


def ulp_diff(a, b):
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    return np.abs(ia - ib)


def render_both(gpu, scene, cam, p, threads=16):
    ref, st_ref = O.render(cam, scene, p, threads)
    gpu.set_scene(scene, cam.time0, cam.time0 + cam.shutter)
    out = {}
    for accel in BOTH:
        p.accel = accel
        out[accel] = gpu.render(cam, p)
    # Since round 2 a BVH request on a scene of <= 48 spheres is routed to the list-walk kernel (RTW_OPT_LIST_WALK_MAX): render
    # once more with the tree FORCED, so that the traversal kernels (all of render_bvh's builds, the GEOM ones included) keep
    # being compared with the oracle on the small scenes most of these tests use.
    gpu.set_option(R.OPT_LIST_WALK_MAX, 0)
    try:
        p.accel = R.ACCEL_BVH
        out["tree forced"] = gpu.render(cam, p)
    finally:
        gpu.set_option(R.OPT_LIST_WALK_MAX, 48)
    return ref, st_ref, out


def test_native_library_is_the_one_loaded(gpu):
    maps = open("/proc/self/maps").read()
    assert "librtw_hip.so" in maps
    assert R.device_count() >= 1


# ---- config 1: the reference's CPU-runnable case, full size -----------------------------------------
def test_c1_full_size_bit_exact(gpu):
    scene = R.Scene.generate(R.SCENE_C1)
    cam, p = R.default_view(R.SCENE_C1)         # 400 x 225, 10 spp, depth 10, seed 1
    p.gamma = 1.0
    ref, st_ref, out = render_both(gpu, scene, cam, p)
    for accel, (img, st) in out.items():
        assert st.segments == st_ref.segments and st.camera_rays == st_ref.camera_rays == 400 * 225 * 10
        assert st.nan_pixels == 0
        assert np.array_equal(img, ref), f"accel {accel}"
    # with the reference's gamma 2: powf is the only difference
    p.gamma = 2.0
    ref, _, out = render_both(gpu, scene, cam, p)
    for accel, (img, _) in out.items():
        assert ulp_diff(img, ref).max() <= 2
        assert np.abs(img - ref).max() < 1e-6


@pytest.mark.parametrize("which", [R.SCENE_METAL_TEST, R.SCENE_C2, R.SCENE_C4])
def test_scenes_bit_exact_small(gpu, which):
    scene, cam, p = small_view(which, 160, 90, 16)
    p.gamma = 1.0
    ref, st_ref, out = render_both(gpu, scene, cam, p)
    for accel, (img, st) in out.items():
        assert st.segments == st_ref.segments
        assert np.array_equal(img, ref), f"accel {accel}: {np.abs(img - ref).max()}"


def test_c5_motion_blur_and_texture(gpu):
    """Moving spheres are exact; the textured ground goes through atan2f/acosf, whose last-bit
    differences between glibc and ocml may move a hit across a texel edge."""
    scene, cam, p = small_view(R.SCENE_C5, 160, 90, 16)
    p.gamma = 1.0
    ref, st_ref, out = render_both(gpu, scene, cam, p)
    for accel, (img, st) in out.items():
        assert st.segments == st_ref.segments          # texel choice never changes the path
        bad = (np.abs(img - ref).max(axis=2) > 0).sum()
        assert bad <= 0.002 * 160 * 90, bad
        assert np.abs(img - ref).max() < 0.05
    assert np.array_equal(out[R.ACCEL_BRUTE][0], out[R.ACCEL_BVH][0])


@pytest.mark.parametrize("sampler", [R.SAMPLER_ROW, R.SAMPLER_STRATIFIED, R.SAMPLER_CENTRES, R.SAMPLER_NO_RAND])
@pytest.mark.parametrize("integrator", [R.INTEGRATOR_GRADIENT, R.INTEGRATOR_NORMAL, R.INTEGRATOR_FLAG])
def test_every_sampler_and_integrator(gpu, sampler, integrator):
    scene, cam, p = small_view(R.SCENE_METAL_TEST, 96, 54, 10)
    cam.lens_radius = 0.02
    p.sampler, p.integrator, p.gamma = sampler, integrator, 1.0
    ref, st_ref, out = render_both(gpu, scene, cam, p)
    for accel, (img, st) in out.items():
        assert st.camera_rays == st_ref.camera_rays and st.segments == st_ref.segments
        assert np.array_equal(img, ref)


def test_bg_color_integrator_with_emission_and_nan_poison(gpu):
    light = R.Sphere.new((0, 3, -1), 1.0, (1, 1, 1), R.SCATTER_M)
    for k in range(3):
        light.pod.emitted[k] = 4.0
    scene = R.Scene([R.Sphere.with_albedo((0, -100.5, -1), 100.0, (0.5, 0.5, 0.5)), R.Sphere.with_albedo((0, 0, -1), 0.5, (0.8, 0.3, 0.3)),
                     R.Sphere.new((1.1, 0, -1), 0.5, (0.9, 0.9, 0.9), R.FUZZY3_M), light], background=(0.05, 0.05, 0.1))
    cam, _ = O.viewport_new(96, np.float32(96) / np.float32(54))
    p = flag_params(depth=8)
    p.width, p.height, p.samples, p.integrator, p.sampler, p.maxt, p.gamma = 96, 54, 16, R.INTEGRATOR_BG_COLOR, R.SAMPLER_ROW, 10000.0, 1.0
    ref, st_ref, out = render_both(gpu, scene, cam, p)
    for accel, (img, st) in out.items():
        assert st.segments == st_ref.segments and st.nan_pixels == st_ref.nan_pixels
        assert np.array_equal(np.isnan(img), np.isnan(ref))
        ok = ~np.isnan(ref)
        assert np.array_equal(img[ok], ref[ok])


def test_seed_changes_image_and_is_reproducible(gpu):
    scene, cam, p = small_view(R.SCENE_C1, 64, 36, 4)
    gpu.set_scene(scene)
    a, _ = gpu.render(cam, p)
    b, _ = gpu.render(cam, p)
    p.seed = 2
    c, _ = gpu.render(cam, p)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    ref, _ = O.render(cam, scene, p)
    assert np.abs(c - ref).max() < 1e-6


# ---- edge cases --------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [0, 1, 2, 3, 17])
def test_tiny_scenes(gpu, n):
    rng = np.random.default_rng(n)
    spheres = [R.Sphere.with_albedo(rng.uniform(-1, 1, 3) + [0, 0, -2], float(rng.uniform(0.2, 0.6)), rng.uniform(0.2, 0.9, 3),
                                    [R.SCATTER_M, R.METALLIC_M, R.GLASS_M][i % 3]) for i in range(n)]
    scene = R.Scene(spheres)
    cam, _ = O.viewport_new(50, np.float32(50) / np.float32(30))
    p = flag_params(depth=5)
    p.width, p.height, p.samples, p.integrator, p.sampler, p.gamma = 50, 30, 4, R.INTEGRATOR_GRADIENT, R.SAMPLER_ROW, 1.0
    ref, st_ref, out = render_both(gpu, scene, cam, p)
    for accel, (img, st) in out.items():
        assert st.segments == st_ref.segments
        assert np.array_equal(img, ref)


@pytest.mark.parametrize("w,h", [(1, 1), (7, 3), (9, 17), (64, 8), (65, 9)])
def test_ragged_image_sizes(gpu, w, h):
    scene = R.Scene.generate(R.SCENE_C1)
    cam, hh = O.viewport_new(w, np.float32(w) / np.float32(h))
    assert hh == h
    p = flag_params(depth=4)
    p.width, p.height, p.samples, p.integrator, p.sampler, p.gamma = w, h, 3, R.INTEGRATOR_GRADIENT, R.SAMPLER_ROW, 1.0
    ref, _, out = render_both(gpu, scene, cam, p)
    for accel, (img, st) in out.items():
        assert img.shape == (h, w, 3) and np.array_equal(img, ref)


@pytest.mark.parametrize("depth", [0, 1, 2])
def test_depth_limits(gpu, depth):
    scene, cam, p = small_view(R.SCENE_C1, 48, 27, 4)
    p.depth, p.gamma = depth, 1.0
    ref, st_ref, out = render_both(gpu, scene, cam, p)
    for accel, (img, st) in out.items():
        assert st.segments == st_ref.segments
        assert np.array_equal(img, ref)
    if depth == 0:
        assert st_ref.segments == 0 and not ref.any()


def test_big_sphere_list_and_equal_radii(gpu):
    """Scenes with several huge spheres (kept outside the tree) and with all-equal radii (none kept outside)."""
    rng = np.random.default_rng(5)
    eq = [R.Sphere.with_albedo(rng.uniform(-3, 3, 3) + [0, 0, -5], 0.4, rng.uniform(0.2, 0.9, 3)) for _ in range(60)]
    big = eq + [R.Sphere.with_albedo((0, -1000.5, -5), 1000.0, (0.5, 0.5, 0.5)), R.Sphere.with_albedo((0, 0, -1040), 1000.0, (0.9, 0.2, 0.2), R.METALLIC_M),
                R.Sphere.with_albedo((1030, 0, -5), 1000.0, (0.2, 0.9, 0.2))]
    for spheres in (eq, big):
        scene = R.Scene(spheres)
        cam, _ = O.viewport_new(80, np.float32(80) / np.float32(45))
        p = flag_params(depth=8)
        p.width, p.height, p.samples, p.integrator, p.sampler, p.gamma, p.maxt = 80, 45, 8, R.INTEGRATOR_GRADIENT, R.SAMPLER_ROW, 1.0, 100000.0
        ref, st_ref, out = render_both(gpu, scene, cam, p)
        for accel, (img, st) in out.items():
            assert st.segments == st_ref.segments
            assert np.array_equal(img, ref)


def test_row_partition_is_image_invariant(gpu):
    """Multi-GPU contract: any (row_block, part_count) split reassembles to the unsplit image bit for bit."""
    scene, cam, p = small_view(R.SCENE_C2, 96, 54, 4)
    gpu.set_scene(scene)
    full, st_full = gpu.render(cam, p)
    for row_block, count in ((8, 2), (8, 8), (1, 3), (16, 4)):
        asm = np.zeros_like(full)
        seg = 0
        for i in range(count):
            p.row_block, p.part_index, p.part_count = row_block, i, count
            rows = [r for r in range(54) if (r // row_block) % count == i]
            img, st = gpu.render(cam, p)
            assert st.rows == len(rows) and img.shape[0] == len(rows)
            asm[rows] = img
            seg += st.segments
        assert np.array_equal(asm, full) and seg == st_full.segments


def test_device_output_pointer(gpu):
    """out_rgb may be device memory of the context's GPU (the bench path: nothing leaves HBM)."""
    import torch
    scene, cam, p = small_view(R.SCENE_C1, 64, 36, 4)
    gpu.set_scene(scene)
    host, _ = gpu.render(cam, p)
    t = torch.zeros((36, 64, 3), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    gpu.render(cam, p, out=t.data_ptr())
    assert np.array_equal(t.cpu().numpy(), host)


def test_error_codes(gpu):
    L = R.lib()
    scene, cam, p = small_view(R.SCENE_C1, 16, 9, 1)
    out = np.zeros((9, 16, 3), np.float32)
    h = C.c_void_p()
    assert L.rtw_ctx_create(0, C.byref(h)) == 0
    assert L.rtw_ctx_render(h, C.byref(cam), C.byref(p), out.ctypes.data_as(C.c_void_p), None) == -6      # NO_SCENE
    assert L.rtw_ctx_set_scene(h, C.byref(scene.pod), 0.0, 0.0) == 0
    bad = R.RtwParams.from_buffer_copy(p)
    bad.integrator = 9
    assert L.rtw_ctx_render(h, C.byref(cam), C.byref(bad), out.ctypes.data_as(C.c_void_p), None) == -1
    bad = R.RtwParams.from_buffer_copy(p)
    bad.samples = 0
    assert L.rtw_ctx_render(h, C.byref(cam), C.byref(bad), out.ctypes.data_as(C.c_void_p), None) == -1
    assert L.rtw_ctx_render(h, C.byref(cam), C.byref(p), None, None) == -1
    assert L.rtw_ctx_render(h, C.byref(cam), C.byref(p), out.ctypes.data_as(C.c_void_p), None) == 0
    L.rtw_ctx_destroy(h)
    assert L.rtw_ctx_create(99, C.byref(h)) == -2


# ---- BASELINE's full sizes: size-independent properties ------------------------------------------------
def test_c2_full_size_bvh_equals_brute_force(gpu):
    """Config 2 (Book-1 final, 1200 x 675, 100 spp, depth 50): the BVH path returns the brute-force
    image bit for bit, the same segment count, and matches the oracle on a strided subset of rows."""
    scene = R.Scene.generate(R.SCENE_C2)
    cam, p = R.default_view(R.SCENE_C2)
    gpu.set_scene(scene)
    p.accel = R.ACCEL_BVH
    a, sa = gpu.render(cam, p)
    p.accel = R.ACCEL_BRUTE
    b, sb = gpu.render(cam, p)
    assert sa.segments == sb.segments and sa.camera_rays == 1200 * 675 * 100
    assert np.array_equal(a, b)
    assert sa.nan_pixels == 0 and np.isfinite(a).all()
    # oracle on every 45th 8-row block (15 blocks of 8 rows = 120 rows ~ 14 M camera rays)
    p.row_block, p.part_index, p.part_count = 8, 0, 6
    ref, st = O.render(cam, scene, p, threads=16)
    rows = [r for r in range(675) if (r // 8) % 6 == 0]
    assert ulp_diff(a[rows], ref).max() <= 2


def test_c3_full_size_properties(gpu):
    """Config 3's per-GPU work at full size (1920 x 1080 x 500 spp): runs, finite, exact sample count,
    and equals the two-way row split rendered separately (partition invariance at scale)."""
    scene = R.Scene.generate(R.SCENE_C2)
    cam, p = R.default_view(R.SCENE_C5)      # 1920 x 1080 x 500, depth 50 framing
    cam.shutter = 0.0
    gpu.set_scene(scene)
    full, st = gpu.render(cam, p)
    assert st.camera_rays == 1920 * 1080 * 500 and st.nan_pixels == 0 and np.isfinite(full).all()
    p.row_block, p.part_index, p.part_count = 8, 1, 2
    half, st_half = gpu.render(cam, p)
    rows = [r for r in range(1080) if (r // 8) % 2 == 1]
    assert np.array_equal(full[rows], half)
    # ... and the list walk renders the same frame, bit for bit (BVH == brute force at the headline size)
    p.row_block, p.part_index, p.part_count = 1, 0, 1
    p.accel = R.ACCEL_BRUTE
    brute, st_b = gpu.render(cam, p)
    assert st_b.segments == st.segments and np.array_equal(full, brute)


def test_rust2_model_bit_exact(gpu):
    """Rust2's ray_color + Material trait objects + Camera/centre sampler on the GPU."""
    from tests.test_oracle_golden import rust2_view
    scene, cam, p = rust2_view(96, 54, 16, 8)
    p.gamma = 1.0
    ref, st_ref, out = render_both(gpu, scene, cam, p)
    for accel, (img, st) in out.items():
        assert st.segments == st_ref.segments and st.camera_rays == st_ref.camera_rays
        assert np.array_equal(img, ref), f"accel {accel}"
    p.depth = 0
    ref, st_ref, out = render_both(gpu, scene, cam, p)
    for accel, (img, st) in out.items():
        assert st.segments == 0 and np.array_equal(img, ref)
    # Book-1 scene through the Rust2 model with the row sampler
    scene, cam, p = small_view(R.SCENE_C2, 96, 54, 8)
    p.integrator, p.gamma = R.INTEGRATOR_RUST2, 1.0
    scene.pod.background[0], scene.pod.background[1], scene.pod.background[2] = 0.7, 0.8, 1.0
    ref, st_ref, out = render_both(gpu, scene, cam, p)
    for accel, (img, st) in out.items():
        assert st.segments == st_ref.segments and np.array_equal(img, ref)


def test_lds_and_global_node_variants_agree(gpu):
    """The f16 LDS-resident node copy (outward-rounded boxes) and the f32 global nodes prune differently
    but must return the same image, bit for bit."""
    for which in (R.SCENE_C2, R.SCENE_C5, R.SCENE_C4):
        scene, cam, p = small_view(which, 160, 90, 8)
        p.accel = R.ACCEL_BVH
        gpu.set_scene(scene, cam.time0, cam.time0 + cam.shutter)
        a, sa = gpu.render(cam, p)
        p.flags = R.FLAG_GLOBAL_NODES
        b, sb = gpu.render(cam, p)
        assert sa.segments == sb.segments and np.array_equal(a, b)
        assert sa.node_tests >= sb.node_tests          # looser boxes can only visit more


def test_render_multi_frame_loop(gpu, rtw):
    """render_multi (viewport.rs:249-269): frame f is rendered at time f / fps; a moving sphere moves between
    frames and every frame equals the oracle's frame with the same time0."""
    moving = R.Sphere.new_moving((0.0, 0.0, -1.5), 0.4, (0.9, 0.4, 0.4), R.SCATTER_M, (0.0, 3.0, 0.0))
    scene = R.Scene([R.Sphere.with_albedo((0, -100.5, -1), 100.0, (0.5, 0.5, 0.5)), moving,
                     R.Sphere.new((1.0, 0.0, -1.5), 0.4, (0.8, 0.8, 0.8), R.METALLIC_M)])
    vp = R.Viewport.new_from_res(64, 36, 8, 6, 1.0)
    vp.fps, vp.shutter_speed, vp.start_frame, vp.number_of_frames = 10.0, 0.05, 1, 3
    video = vp.render_multi(R.INTEGRATOR_GRADIENT, scene)
    assert len(video) == 3 and not np.array_equal(video[0], video[2])
    for k, frame in enumerate(video):
        vp.frame = 1 + k
        ref, _ = O.render(vp.camera(), scene, vp.params(R.INTEGRATOR_GRADIENT, R.SAMPLER_ROW, R.ACCEL_BRUTE))
        assert np.array_equal(frame, ref)


@pytest.mark.parametrize("which", [R.SCENE_C4, R.SCENE_C5])
def test_c4_c5_full_size(gpu, which):
    """Configs 4 (dielectric-heavy, 1920x1080x1000 spp) and 5 (motion blur + image-textured ground,
    1920x1080x500 spp) at BASELINE's full size: BVH == brute force bit for bit, exact sample count, finite;
    one 8-row block is also checked against the oracle."""
    scene = R.Scene.generate(which)
    cam, p = R.default_view(which)
    gpu.set_scene(scene, cam.time0, cam.time0 + cam.shutter)
    p.accel = R.ACCEL_BVH
    a, sa = gpu.render(cam, p)
    p.accel = R.ACCEL_BRUTE
    b, sb = gpu.render(cam, p)
    assert sa.camera_rays == sb.camera_rays == 1920 * 1080 * p.samples
    assert sa.segments == sb.segments and sa.nan_pixels == 0 and np.isfinite(a).all()
    assert np.array_equal(a, b)
    print(f"\\nconfig {which}: bvh {sa.kernel_ms:.1f} ms ({sa.segments / sa.kernel_ms / 1e6:.2f} Gseg/s), brute {sb.kernel_ms:.1f} ms, "
          f"{sa.segments / sa.camera_rays:.2f} segments per camera ray")
    p.row_block, p.part_index, p.part_count = 8, 67, 135          # the 8 rows in the middle of the frame
    ref, st = O.render(cam, scene, p, threads=16)
    rows = list(range(67 * 8, 67 * 8 + 8))
    if which == R.SCENE_C5:     # texel edges: atan2f/acosf may differ in the last bit between glibc and ocml
        assert (np.abs(a[rows] - ref).max(axis=2) > 1e-6).sum() <= 0.002 * 8 * 1920
    else:
        assert ulp_diff(a[rows], ref).max() <= 2


def test_sample_bank_bands_and_chunk_lengths(rtw):
    """The per-sample radiance bank may be rendered in bands of tile rows (memory budget) and with any chunk
    length (samples per work unit): neither changes a bit of the image."""
    scene, cam, p = small_view(R.SCENE_C2, 200, 120, 37)
    p.gamma, p.accel = 1.0, R.ACCEL_BVH
    ref, st_ref = O.render(cam, scene, p)
    with rtw.Renderer(0) as gpu:                             # own context: the options below must not leak into other tests
        gpu.set_scene(scene)
        base, st = gpu.render(cam, p)
        assert np.array_equal(base, ref) and st.segments == st_ref.segments
        for chunk in (1, 5, 37, 64):
            gpu.set_option(R.OPT_CHUNK_LEN, chunk)
            img, st = gpu.render(cam, p)
            assert np.array_equal(img, ref) and st.camera_rays == 200 * 120 * 37, chunk
        gpu.set_option(R.OPT_CHUNK_LEN, 0)
        gpu.set_option(R.OPT_SAMPLE_BANK_GB, 0.003)          # ~3 MiB: forces several bands (one tile row is 25*10*64*4*12 B = 0.73 MiB)
        img, st = gpu.render(cam, p)
        assert np.array_equal(img, ref) and st.segments == st_ref.segments
        gpu.set_option(R.OPT_SAMPLE_BANK_GB, 0.0005)         # less than one tile row: refused, not truncated
        with pytest.raises(R.RtwError) as e:
            gpu.render(cam, p)
        assert e.value.status == -4
        with pytest.raises(R.RtwError):                      # unknown key / out-of-range value
            gpu.set_option(99, 1)
        with pytest.raises(R.RtwError):
            gpu.set_option(R.OPT_CHUNK_LEN, -1)


def test_example_program_through_the_c_abi(gpu, tmp_path):
    """examples/render_scene.cpp (C ABI only, g++) renders metal_test like the reference's test function and its
    PNG equals the Python path's quantised image; it also round-trips the scene through the JSON wire format."""
    import os, struct, subprocess, zlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "render_scene")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(root, "raytracing-in-a-weekend_amd", "csrc"), "example"], check=True, capture_output=True)
    png, js = str(tmp_path / "m.png"), str(tmp_path / "m.json")
    out = subprocess.run([exe, "--scene", "metal", "--out", png, "--dump-json", js], check=True, capture_output=True, text=True).stdout
    assert "4 spheres + 0 quads + 0 instances, 400x225, 9000000 camera rays" in out
    scene = R.Scene.generate(R.SCENE_METAL_TEST)
    cam, p = R.default_view(R.SCENE_METAL_TEST)
    gpu.set_scene(scene)
    want = R.quantize_u8(gpu.render(cam, p)[0])

    def read_png(path):
        raw = open(path, "rb").read()
        pos, idat, hdr = 8, b"", None
        while pos < len(raw):
            n, t = struct.unpack(">I4s", raw[pos:pos + 8])
            if t == b"IHDR": hdr = struct.unpack(">II", raw[pos + 8:pos + 16])
            if t == b"IDAT": idat += raw[pos + 8:pos + 8 + n]
            pos += 12 + n
        w, h = hdr
        return np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)[:, 1:].reshape(h, w, 3)
    assert np.array_equal(read_png(png), want)
    png2 = str(tmp_path / "m2.png")
    subprocess.run([exe, "--json", js, "--width", "400", "--height", "225", "--spp", "100", "--depth", "10", "--out", png2], check=True, capture_output=True)
    # the JSON scene renders with the default view's ROW sampler at maxt 1e5 unless told otherwise: same scene, so only sanity here
    assert read_png(png2).shape == (225, 400, 3)
    # presentation_image through the C ABI alone (Scene::new with quads and instances), reduced spp
    png3 = str(tmp_path / "p.png")
    out = subprocess.run([exe, "--scene", "presentation", "--spp", "64", "--out", png3], check=True, capture_output=True, text=True).stdout
    assert "1 spheres + 6 quads + 2 instances, 400x400, 10240000 camera rays" in out
    scene = R.Scene.generate_geom(R.SCENE_PRESENTATION)
    cam, p = R.default_view(R.SCENE_PRESENTATION)
    p.samples = 64
    gpu.set_scene(scene)
    assert np.array_equal(read_png(png3), R.quantize_u8(gpu.render(cam, p)[0]))


def test_bench_two_ranks_rehearsal_on_one_gpu(gpu):
    """bench.py's N = 2 path end to end (partition, per-rank render, gather, counter reduction, JSON line) with
    both ranks on this one GPU and gloo standing in for RCCL; the 8-GPU run itself is the driver's."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, RTW_BENCH_BACKEND="gloo", RTW_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--spp", "8"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert abs(d["roofline"]["units_per_launch"]["segments"] * 2 / (1920 * 1080 * 8) - d["config"]["segments_per_camera_ray"]) < 1e-3


# ---- quads, instances, constant-density medium -------------------------------------------
def _blocks16(img):
    q = np.round(np.clip(np.nan_to_num(img.astype(np.float64)) * 255.0, 0, 255))
    h, w, _ = q.shape
    return q.reshape(h // 16, 16, w // 16, 16, 3).mean(axis=(1, 3))


def test_quad_test_scene_bit_exact(gpu):
    # objects/quad.rs:152-299 at its own size: 400 x 400, 100 spp (stratified), depth 10
    scene = R.Scene.generate_geom(R.SCENE_QUAD_TEST)
    cam, p = R.default_view(R.SCENE_QUAD_TEST)
    p.gamma = 1.0
    ref, st_ref, out = render_both(gpu, scene, cam, p)
    for accel, (img, st) in out.items():
        assert st.segments == st_ref.segments and st.quad_tests == st_ref.quad_tests == 5 * st.segments
        assert np.array_equal(img, ref), f"accel {accel}"


def test_presentation_scene_bit_exact_and_matches_the_reference_png(gpu):
    # presentation_image (main.rs:89-419): sphere + 6 quads (mirror, light, lambert) + smoke box + glass pane
    scene = R.Scene.generate_geom(R.SCENE_PRESENTATION)
    cam, p = R.default_view(R.SCENE_PRESENTATION)
    p.gamma, p.samples = 1.0, 24
    ref, st_ref, out = render_both(gpu, scene, cam, p)
    for accel, (img, st) in out.items():
        assert st.segments == st_ref.segments and st.quad_tests == st_ref.quad_tests and st.sphere_tests == st_ref.sphere_tests
        assert st.nan_pixels == st_ref.nan_pixels
        assert np.array_equal(img, ref, equal_nan=True), f"accel {accel}"
    # the frame as the reference rendered it (400 x 400 x 2500 spp, gamma 2) against the reference's own PNG
    cam, p = R.default_view(R.SCENE_PRESENTATION)
    img, st = gpu.render(cam, p)
    assert st.camera_rays == 400 * 400 * 2500
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_images.npz"))
    d = _blocks16(img) - gold["presentation_blocks16"]
    assert np.abs(d).mean() < 0.35 and np.abs(d).max() < 3.0 and np.abs(d.mean(axis=(0, 1))).max() < 0.15, (np.abs(d).mean(), np.abs(d).max())
    assert 5 <= st.nan_pixels <= 80          # the reference's PNG has 27 such black pixels


def test_first_frame_scene_bit_exact_and_matches_the_reference_png(gpu):
    scene = R.Scene.generate(R.SCENE_FIRST_FRAME)
    cam, p = R.default_view(R.SCENE_FIRST_FRAME)
    p.gamma = 1.0
    ref, st_ref, out = render_both(gpu, scene, cam, p)
    for accel, (img, st) in out.items():
        assert st.segments == st_ref.segments
        assert np.array_equal(img, ref), f"accel {accel}"
    cam, p = R.default_view(R.SCENE_FIRST_FRAME)
    p.samples = 1600                            # 16x the reference's 100 spp: the comparison is limited by ITS noise
    img, _ = gpu.render(cam, p)
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_images.npz"))
    d = _blocks16(img) - gold["first_frame_blocks16"]
    assert np.abs(d).mean() < 0.3 and np.abs(d).max() < 3.0 and np.abs(d.mean(axis=(0, 1))).max() < 0.15, (np.abs(d).mean(), np.abs(d).max())


def test_mixed_scene_instances_with_spheres_textures_motion(gpu):
    # everything at once: top-level spheres (one moving, one image-textured) + quads (one image-textured, one light) +
    # an instance holding spheres and quads under a 3-axis rotation + a smoke sphere-and-box instance
    rng = np.random.default_rng(11)
    tex = rng.uniform(0.1, 0.9, size=(3, 5, 3)).astype(np.float32)
    spheres = [R.Sphere.new((0, -100.5, -1), 100.0, (0.8, 0.8, 0.0), R.SCATTER_M),
               R.Sphere.new_moving((-1.2, 0.0, -1.5), 0.4, (0.9, 0.4, 0.4), R.FUZZY3_M, (0.0, 3.0, 0.0)),
               R.Sphere.new_with_texture((1.3, 0.1, -1.8), 0.5, None, R.SCATTER_M, 0)]
    quads = [R.Quad.new((-3, -0.5, -4), (6, 0, 0), (0, 3, 0), R.SCATTER_M, (1, 1, 1), tex_index=0),
             R.Quad.new((-0.5, 2.0, -2.5), (1, 0, 0), (0, 0, 1), (0.0, 0.0, 1.0), (1, 1, 1), emitted=(6, 5, 4))]
    a = R.Instance.new([R.Sphere.new((0.0, 0.0, 0.0), 0.3, (0.5, 0.7, 0.9), R.GLASS_M), R.Sphere.new((0.5, 0.2, 0.1), 0.2, (0.9, 0.9, 0.9), R.METALLIC_M)],
                       [R.Quad.new((-0.6, -0.4, 0.4), (1.2, 0, 0), (0, 0.8, 0), R.METALLIC_M, (0.8, 0.8, 0.8))])
    a.rotate((0.3, -0.7, 1.1)); a.translate((0.2, 0.3, -1.2))
    b = R.Instance.new_box((-0.4, -0.3, -0.3), (0.4, 0.3, 0.3), (0.3, 0.3, 0.3), R.SCATTER_M)
    b.spheres.append(R.Sphere.new((0.0, 0.5, 0.0), 0.25, (0.6, 0.6, 0.6), R.SCATTER_M).pod)
    b.rotate((0.0, 0.6, 0.0)); b.translate((-0.4, 0.0, -0.9)); b.const_density(3.0)
    scene = R.Scene(spheres, textures=[tex], background=(0.05, 0.06, 0.1), quads=quads, instances=[a, b])
    vp = R.Viewport.new_from_res(160, 96, 16, 12, 1.0, vfov=70.0, lens_radius=0.02)
    vp.shutter_speed, vp.fps = 1.0 / 30.0, 30.0
    cam = vp.camera()
    for integrator in (R.INTEGRATOR_BG_COLOR, R.INTEGRATOR_GRADIENT, R.INTEGRATOR_RUST2, R.INTEGRATOR_NORMAL):
        p = vp.params(integrator, R.SAMPLER_ROW)
        ref, st_ref, out = render_both(gpu, scene, cam, p)
        for accel, (img, st) in out.items():
            assert st.segments == st_ref.segments and st.quad_tests == st_ref.quad_tests, (integrator, accel)
            same = np.isclose(img, ref, rtol=0, atol=0, equal_nan=True).all(axis=2)
            # sphere / quad image textures go through atan2f / acosf / floorf of a product: a texel edge may fall differently
            assert same.mean() > 0.999, (integrator, accel, same.mean())
            assert np.nanmax(np.abs(np.where(np.isnan(ref), 0, img - ref))) < 1.0


# ---- exponent-range edges: the guarded fast paths (unit(): shared reciprocal, unscaled sqrt) must agree with IEEE everywhere
@pytest.mark.parametrize("log2_scale", [0, -70, 70, -126])
def test_direction_scale_and_zero_components(gpu, log2_scale):
    """Ray directions with exactly-zero components (the centre row / column of a power-of-two camera) and directions scaled by
    2^-70 / 2^70 / 2^-126 (denormal components): every lane outside [2^-40, 2^40] sends its wave down the generic IEEE expansion,
    and the image must stay bit-identical to the oracle either way.  At 2^70 |d|^2 overflows f32: the reference's quadratic then
    reports NaN roots as hits, and the BVH kernel has to fall back to the list walk to reproduce that.  (The reference never normalises directions, so a scaled
    camera is a legal input; scaling by a power of two scales every t exactly.)"""
    scene = R.Scene.generate(R.SCENE_METAL_TEST)
    sc = float(np.ldexp(1.0, log2_scale))
    cam = R.RtwCamera()
    for k, v in enumerate((-1.0 * sc, 0.5 * sc, -1.0 * sc)):
        cam.pixel00[k] = v                               # pixel (8, 4) looks straight down -z: x == 0 and y == 0 exactly
    cam.delta_u[0], cam.delta_v[1] = 0.125 * sc, -0.125 * sc
    cam.u[0], cam.v[1] = 1.0, 1.0
    vp = R.Viewport(cam, 16, 8, 32, 12, 1.0)
    vp.mint, vp.maxt = 0.001 / sc if log2_scale > -100 else 0.001, min(1000.0 / sc, 3e38)
    if log2_scale == -126:
        vp.mint, vp.maxt = 1e30, 3e38                    # t = distance / |d| is ~1e38 here
    for sampler in (R.SAMPLER_NO_RAND, R.SAMPLER_ROW):
        p = vp.params(R.INTEGRATOR_GRADIENT, sampler)
        ref, st_ref, out = render_both(gpu, scene, vp.camera(), p)
        assert st_ref.segments > st_ref.camera_rays or log2_scale == -126      # something was hit and scattered
        for accel, (img, st) in out.items():
            assert st.segments == st_ref.segments, (log2_scale, sampler, accel)
            assert np.array_equal(img, ref, equal_nan=True), (log2_scale, sampler, accel)


# ---- BVH stress: the conservative traversal must return the brute-force hit for adversarial geometry ------------------------
@pytest.mark.parametrize("kind", ["dense_large", "radius_spread", "coincident_tangent", "moving_swarm"])
def test_bvh_equals_brute_force_on_adversarial_scenes(gpu, kind):
    """BVH vs the list walk (and vs the oracle), bit for bit, on scenes built to stress the pruning argument: thousands of
    spheres (f32 nodes in global memory, deep tree), radii spread over 5 decades, exactly coincident / tangent / nested spheres
    (ties in t must go to the lower index), and a swarm of fast movers whose bounds are time-expanded."""
    rng = np.random.default_rng({"dense_large": 1, "radius_spread": 2, "coincident_tangent": 3, "moving_swarm": 4}[kind])
    mats = [R.SCATTER_M, R.METALLIC_M, R.GLASS_M, R.FUZZY3_M, R.GLASSR_M]
    spheres = []
    shutter = 0.0
    if kind == "dense_large":
        for i in range(3000):
            spheres.append(R.Sphere.with_albedo(rng.uniform(-12, 12, 3) + [0, 0, -14], float(rng.uniform(0.05, 0.5)), rng.uniform(0.2, 0.95, 3), mats[i % 5]))
    elif kind == "radius_spread":
        for i in range(700):
            r = float(10.0 ** rng.uniform(-3.5, 1.2))
            spheres.append(R.Sphere.with_albedo(rng.uniform(-6, 6, 3) * (1 + r) + [0, 0, -10 - 2 * r], r, rng.uniform(0.2, 0.95, 3), mats[i % 5]))
    elif kind == "coincident_tangent":
        for i in range(40):
            c = np.round(rng.uniform(-3, 3, 3) * 4) / 4 + [0, 0, -6]
            r = float(rng.choice([0.25, 0.5]))
            spheres.append(R.Sphere.with_albedo(c, r, rng.uniform(0.2, 0.95, 3), mats[i % 5]))
            spheres.append(R.Sphere.with_albedo(c, r, rng.uniform(0.2, 0.95, 3), mats[(i + 1) % 5]))                  # exactly coincident
            spheres.append(R.Sphere.with_albedo(c + [2 * r, 0, 0], r, rng.uniform(0.2, 0.95, 3), mats[(i + 2) % 5]))   # tangent
            spheres.append(R.Sphere.with_albedo(c, r / 2, rng.uniform(0.2, 0.95, 3), mats[(i + 3) % 5]))              # nested, concentric
    else:
        shutter = 1.0 / 30.0
        for i in range(600):
            s = R.Sphere.with_albedo(rng.uniform(-5, 5, 3) + [0, 0, -8], float(rng.uniform(0.05, 0.3)), rng.uniform(0.2, 0.95, 3), mats[i % 5],
                                     velocity=tuple(rng.uniform(-60, 60, 3)))                               # up to 2 units per shutter
            spheres.append(s)
    scene = R.Scene(spheres)
    vp = R.Viewport.new_from_res(128, 72, 8, 10, 1.0, vfov=70.0, lens_radius=0.01)
    vp.shutter_speed, vp.fps = shutter, 30.0
    cam = vp.camera()
    p = vp.params(R.INTEGRATOR_GRADIENT, R.SAMPLER_ROW)
    ref, st_ref, out = render_both(gpu, scene, cam, p)
    for accel, (img, st) in out.items():
        assert st.segments == st_ref.segments, (kind, accel)
        assert np.array_equal(img, ref), (kind, accel, int((img != ref).any(axis=2).sum()))
    assert out[R.ACCEL_BVH][1].node_tests > 0


def test_reference_texture_reflection_scene(gpu):
    """reflection_test of Rust/src/viewport/texture_test.rs:87-138 -- a mirror sphere in front of a huge image-textured mirror
    sphere, 400 x 300, 100 spp (stratified), depth 10 -- with the reference's own asset Rust/assets/squares.png as the image
    (its earthmap.jpg is 1.5 MB; texels as ImageTexture::from_path builds them: u8 / 255).  The texel choice goes through
    atan2f / acosf, so a texel edge may fall differently between glibc and ocml: a handful of pixels, nothing else."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_images.npz"))
    tex = z["squares_png_rgb"].astype(np.float32) / np.float32(255.0)            # [64][128][3]
    assert tex.shape == (64, 128, 3)
    spheres = [R.Sphere.new((0.520, 0.0, -1.0), 0.45, (0.95, 0.95, 0.95), R.METALLIC_M),
               R.Sphere.new_with_texture((-1001.0, 0.0, 0.0), 1000.0, None, R.METALLIC_M, 0)]
    scene = R.Scene(spheres, textures=[tex])
    vp = R.Viewport.new_from_res(400, 300, 100, 10, 2.0, vfov=90.0, origin=(0.0, 0.0, 0.0))
    vp.maxt = 1000.0
    cam = vp.camera()
    p = vp.params(R.INTEGRATOR_GRADIENT, R.SAMPLER_STRATIFIED)
    ref, st_ref, out = render_both(gpu, scene, cam, p)
    assert len(np.unique(np.round(ref.reshape(-1, 3), 2), axis=0)) > 50        # the textured wall and its reflection are in view
    for accel, (img, st) in out.items():
        assert st.segments == st_ref.segments
        same = (ulp_diff(img, ref) <= 2).all(axis=2)
        assert same.mean() > 0.999, (accel, same.mean())
    assert np.array_equal(out[R.ACCEL_BRUTE][0], out[R.ACCEL_BVH][0])
