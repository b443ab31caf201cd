"""Pins the CPU oracle (oracle/rtw_oracle.c) against everything the reference tree holds for this path:
the Rust/cerr trace, the s_test images and per-hit vectors of the reference's own C++ objects, the
reference Camera, and analytic known answers.  CPU only."""
import ctypes as C
import hashlib
import json
import math
import os

import numpy as np
import pytest

import rtw_amd as R
from tests import oracle_binding as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def flag_params(depth=10, flags=R.RtwParams().flags):
    p = R.RtwParams()
    p.width, p.height, p.samples, p.depth = 10, 10, 1, depth
    p.gamma, p.mint, p.maxt = 2.0, 0.001, 1000.0          # glass_tests.rs:13-14
    p.integrator, p.sampler, p.accel, p.flags, p.seed = R.INTEGRATOR_FLAG, R.SAMPLER_NO_RAND, R.ACCEL_BRUTE, flags, 1
    p.row_block, p.part_index, p.part_count = 1, 0, 1
    return p


def glass_scene(mat):
    # glass_tests.rs:102-123 / tests.cpp:277-284: sphere (0,0,-1) r .5 + ground (0,-100.5,-1) r 100 EMPTY_M
    return R.Scene([R.Sphere.new((0.0, 0.0, -1.0), 0.5, (1.0, 1.0, 1.0), mat),
                    R.Sphere.new((0.0, -100.5, -1.0), 100.0, (1.0, 1.0, 1.0), R.EMPTY_M)])


# ---- 1. Rust/cerr --------------------------------------------------------------------------------
def test_cerr_trace_58_pixels():
    data = json.load(open(os.path.join(GOLD, "cerr_trace.json")))
    scene = glass_scene(R.GLASS_M)
    p = flag_params(flags=2)      # RTW_FLAG_CPP_DIELECTRIC: the trace comes from the deterministic C++ dielectric
    assert len(data["pixels"]) == 58
    checked = 0
    for px in data["pixels"]:
        u, v = np.float32(px["x"]) / np.float32(9), np.float32(9 - px["y"]) / np.float32(9)
        assert abs(u - px["u"]) < 1e-6 and abs(v - px["v"]) < 1e-6
        d = (np.float32(-1) + np.float32(2) * u, np.float32(-1) + np.float32(2) * v, np.float32(-1))
        bounces, _ = O.trace_ray((0, 0, 0), d, 0.0, scene, p)
        want = px["bounces"]
        assert len(bounces) == len(want), (px["x"], px["y"], len(bounces), len(want))
        for b, w in zip(bounces, want):
            if w["kind"] == "sky":
                assert b.hit == 0
            elif w["kind"] == "scatter_hit":
                assert b.hit == 1 and b.sphere == 1
            else:
                assert b.hit == 1 and b.sphere == 0
                assert b.front_face == w["front_face"] and (1 - b.cannot_refract) == w["can_refract"]
                assert abs(b.ratio - w["ratio"]) < 2e-6
                facing = np.array(b.normal) * (1.0 if b.front_face else -1.0)
                # the trace prints 6 significant digits (and its C++ build used double intermediates)
                tol = dict(rtol=6e-6, atol=1.5e-6)
                np.testing.assert_allclose(facing, w["facing_normal"], **tol)
                if "unit_dir" in w:
                    np.testing.assert_allclose(np.array(b.unit_dir), w["unit_dir"], **tol)
                np.testing.assert_allclose(np.array(b.point), w["next_origin"], **tol)
                np.testing.assert_allclose(np.array(b.next_dir), w["next_dir"], **tol)
                checked += 1
    assert checked == 36


# ---- 2. s_test images (the reference's own parity definition: identical 8-bit images, RNG-free) ----
@pytest.mark.parametrize("name,mat,flags", [("control", R.METALLIC_M, 0), ("glass", R.GLASS_M, 2)])
def test_s_test_images(name, mat, flags):
    z = np.load(os.path.join(GOLD, "s_test.npz"))
    h, w = [int(x) for x in z[name + "_shape"]]
    want = np.unpackbits(z[name + "_bits"])[: h * w].reshape(h, w).astype(bool)
    scene = R.Scene([R.Sphere.new((0.0, 0.0, -1.0), 0.5, (1.0, 1.0, 1.0), mat),
                     R.Sphere.with_albedo((0.0, -100.5, -1.0), 100.0, (0.8, 0.5, 1.0), R.SCATTER_M)])
    cam, hh = O.viewport_new(300, 1.5)           # Viewport() default camera: Camera(300, 1.5, 90) viewport.h:76
    assert hh == h
    p = flag_params(flags=flags)
    p.width, p.height = w, h
    img, _ = O.render(cam, scene, p, threads=4)
    q = (255.0 * img.astype(np.float64)).astype(np.int32)         # RGB_int: static_cast<int>(255 * c) RGB.cpp:16-20
    yellow = (q == np.array([255, 255, 0])).all(axis=2)
    blue = (q == np.array([0, 0, 255])).all(axis=2)
    assert (yellow | blue).all()
    mism = int((yellow != want).sum())
    assert mism == 0, f"{mism} of {h * w} pixels differ from the reference's {name} image"


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (only exists where /root/reference does)")
def test_ref_s_test_md5():
    """The restated pixel loop of oracle/ref_driver.cpp reproduces the md5 the real viewport.cpp produced."""
    z = np.load(os.path.join(GOLD, "s_test.npz"))
    L = O.ref()
    pa, pb, la, lb = C.POINTER(C.c_char)(), C.POINTER(C.c_char)(), C.c_size_t(), C.c_size_t()
    L.rtw_ref_s_test(C.byref(pa), C.byref(la), C.byref(pb), C.byref(lb))
    assert hashlib.md5(C.string_at(pa, la.value)).hexdigest() == "c3cad547adfa8b69cb9c02308e1ccc3c" == bytes(z["control_md5"]).decode()
    assert hashlib.md5(C.string_at(pb, lb.value)).hexdigest() == "b5ce407b2062997eb6df78452e7249dc" == bytes(z["glass_md5"]).decode()
    L.rtw_ref_free(pa)
    L.rtw_ref_free(pb)


# ---- 3. per-hit vectors of the reference's Sphere::collisionNormal / Material::onHit -----------------
def test_ref_sphere_hits():
    data = json.load(open(os.path.join(GOLD, "ref_sphere_hits.json")))
    p = flag_params(depth=1, flags=2)
    n_hit, errs = 0, []
    for c in data["cases"]:
        scene = R.Scene([R.Sphere.new(c["centre"], c["radius"], (1.0, 1.0, 1.0), c["mat3"])])
        bounces, _ = O.trace_ray(c["origin"], c["dir"], 0.0, scene, p)
        assert len(bounces) == 1
        b = bounces[0]
        assert b.hit == c["hit"], c
        if not c["hit"]:
            continue
        n_hit += 1
        # The C++ twin carries double intermediates (vec3.h:68 dot returns double), the oracle is f32 like
        # the Rust: agreement is ~1 ulp on well-conditioned hits and degrades with the cancellation in
        # b*b - a*c on grazing ones, so bound every case loosely and the bulk tightly.
        got = np.array(b.next_dir, np.float64)
        want = np.array(c["next_dir"], np.float64)
        errs.append((abs(b.t - c["t"]) / max(1.0, abs(c["t"])),
                     np.abs(np.array(b.normal) - c["normal"]).max(),
                     np.abs(np.array(b.point) - c["point"]).max(),
                     # the C++ mirror normalises its direction (materials.cpp:10-13), Rust does not: compare directions
                     np.abs(got / np.linalg.norm(got) - want / np.linalg.norm(want)).max()))
    e = np.array(errs)
    assert (e.max(axis=0) < [4e-6, 2e-4, 4e-5, 5e-4]).all(), e.max(axis=0)
    assert (np.percentile(e, 90, axis=0) < [1e-6, 4e-6, 4e-6, 8e-6]).all(), np.percentile(e, 90, axis=0)
    assert (np.median(e, axis=0) < 6e-7).all(), np.median(e, axis=0)
    assert n_hit == 180


# ---- 4. camera ---------------------------------------------------------------------------------------
def test_ref_camera_and_host_mirror():
    data = json.load(open(os.path.join(GOLD, "ref_camera.json")))
    for c in data["cameras"]:
        cam, h = O.viewport_new(c["width"], np.float32(c["aspect"]), c["vfov"], c["origin"], c["direction"], c["vup"], c["lens_radius"])
        assert h == c["height"]
        for f in ("origin", "u", "v", "pixel00", "delta_u", "delta_v"):
            want = np.array(c[f])
            if f == "pixel00":      # the C++ corner includes the origin (viewport.h:44), Rust's is a direction (viewport.rs:359)
                want = want - np.array(c["origin"])
            np.testing.assert_allclose(np.array(getattr(cam, f)), want, rtol=2e-6, atol=1e-6, err_msg=f)
        # the product's host mirror (librtw_hip.so rtw_viewport_new) is bit-identical to the oracle's restatement
        vp = R.Viewport.new(c["width"], np.float32(c["aspect"]), 1, 1, 2.0, c["vfov"], c["origin"], c["direction"], c["vup"], None, c["lens_radius"])
        assert vp.height == h
        assert bytes(vp.cam) == bytes(cam)


def test_viewport_new_analytic():
    # 1920x1080, vfov 20, default direction: viewport 2 tan(10 deg) tall, pixel (0,0) centre
    cam, h = O.viewport_new(1920, np.float32(1920) / np.float32(1080), 20.0)
    assert h == 1080
    th = 2 * math.tan(math.radians(10))
    tw = th * 1920 / 1080
    np.testing.assert_allclose(np.array(cam.delta_u), [tw / 1920, 0, 0], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(np.array(cam.delta_v), [0, -th / 1080, 0], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(np.array(cam.pixel00), [-tw / 2 + tw / 3840, th / 2 - th / 2160, -1], rtol=1e-6)
    # heights the configs rely on (viewport.rs:346 truncation)
    for w, hh in ((400, 225), (1200, 675), (1920, 1080), (300, 200)):
        assert O.viewport_new(w, np.float32(w) / np.float32(hh))[1] == hh
    # a non-unit direction scales v (w = -dir is not normalised, viewport.rs:342-344)
    cam2, _ = O.viewport_new(100, 1.0, 90.0, None, (0.0, 0.0, -3.0))
    np.testing.assert_allclose(np.array(cam2.v), [0, 3, 0], atol=1e-6)


# ---- 5. RNG known answers ----------------------------------------------------------------------------
def _py_rng(seed, pixel, sample, n):
    M = 0xFFFFFFFF

    def mix(x):
        x ^= x >> 16; x = (x * 0x7feb352d) & M; x ^= x >> 15; x = (x * 0x846ca68b) & M; x ^= x >> 16
        return x
    h = mix(((seed & M) + 0x9E3779B9) & M)
    h = mix(h ^ (seed >> 32)); h = mix(h ^ pixel); h = mix(h ^ sample)
    state, inc = h, mix(h ^ 0x85EBCA6B) | 1
    out = []
    for _ in range(n):
        old = state
        state = (old * 747796405 + inc) & M
        out.append(np.float32(old >> 8) * np.float32(2.0 ** -24))          # the top 24 bits of the LCG state
    return h, inc, out


@pytest.mark.parametrize("seed,pixel,sample,state,inc,first", [
    (1, 0, 0, 3555472974, 503772033, 0.8278230428695679),
    (1, 12345, 7, 1103375227, 3231797855, 0.25689953565597534),
    (0xDEADBEEFCAFEF00D, 89999, 499, 2916675818, 1961906125, 0.6790914535522461)])
def test_rng_known_answers(seed, pixel, sample, state, inc, first):
    st = (C.c_uint32 * 2)()
    O.lib().rtw_oracle_rng_seed(seed, pixel, sample, st)
    assert (st[0], st[1]) == (state, inc)
    h, i, want = _py_rng(seed, pixel, sample, 64)
    assert (h, i) == (state, inc)
    got = [O.lib().rtw_oracle_rng_next(st) for _ in range(64)]
    assert got == [float(x) for x in want]
    assert got[0] == first
    assert all(0.0 <= x < 1.0 for x in got)


# ---- 6. analytic known answers for the material branch -----------------------------------------------
def test_dielectric_normal_incidence_and_tir():
    p = flag_params(depth=4, flags=2)
    scene = glass_scene(R.GLASS_M)
    # straight through the centre: enters and leaves undeviated, ratio 1/1.5 then 1.5
    b, _ = O.trace_ray((0, 0, 0), (0, 0, -1), 0.0, scene, p)
    assert [x.hit for x in b[:2]] == [1, 1]
    assert abs(b[0].t - 0.5) < 1e-6 and b[0].front_face == 1 and abs(b[0].ratio - 1 / 1.5) < 1e-7
    np.testing.assert_allclose(np.array(b[0].next_dir), [0, 0, -1], atol=1e-6)
    assert b[1].front_face == 0 and abs(b[1].ratio - 1.5) < 1e-7 and abs(b[1].t - 1.0) < 1e-6
    np.testing.assert_allclose(np.array(b[1].next_dir), [0, 0, -1], atol=1e-6)
    # from inside, grazing: 1.5 * sin(theta) > 1 -> cannot refract -> mirror reflection
    b, _ = O.trace_ray((0.0, 0.0, -1.0), (1.0, 0.0, 0.05), 0.0, R.Scene([R.Sphere.new((0.0, -0.45, -1.0), 0.5, None, R.GLASS_M)]), p)
    assert b[0].hit == 1 and b[0].front_face == 0 and b[0].cannot_refract == 1
    n = -np.array(b[0].normal)
    ud = np.array(b[0].unit_dir)
    np.testing.assert_allclose(np.array(b[0].next_dir), ud - 2 * np.dot(ud, n) * n, atol=2e-6)


def test_schlick_branch_statistics():
    """With the Rust dielectric (Schlick on), the reflect probability at normal incidence is r0 = 0.04."""
    scene = R.Scene([R.Sphere.new((0.0, 0.0, -1.0), 0.5, None, R.GLASS_M)])
    p = flag_params(depth=1, flags=0)
    refl = 0
    n = 4000
    for s in range(n):
        b, _ = O.trace_ray((0, 0, 0), (0, 0, -1), 0.0, scene, p, pixel=3, sample=s)
        refl += b[0].next_dir[2] > 0
    assert abs(refl / n - 0.04) < 4 * math.sqrt(0.04 * 0.96 / n)


def test_sphere_new_albedo_quirk():
    """Sphere::new(c) reports c*c (col_mod * 1x1 texture, sphere.rs:145,151-173): mirror of white sky."""
    s = R.Sphere.new((0, 0, -1), 0.5, (0.8, 0.5, 1.0), R.METALLIC_M)
    assert list(s.pod.col_mod) == list(s.pod.tex_color)
    p = flag_params(depth=5)
    p.integrator = R.INTEGRATOR_GRADIENT
    _, rgb = O.trace_ray((0, 0, 0), (0, 0, -1), 0.0, R.Scene([s]), p)
    # reflected straight back: sky(0,0,1) = (0.75, 0.85, 1.0), times c*c
    np.testing.assert_allclose(rgb, np.float32([0.75, 0.85, 1.0]) * np.float32([0.8, 0.5, 1.0]) ** 2, rtol=1e-6)


# ---- 7. oracle self-consistency ------------------------------------------------------------------------
def small_view(which, w=64, h=36, samples=4):
    scene = R.Scene.generate(which)
    cam, p = R.default_view(which)
    f = p.width / w
    for k in range(3):
        cam.pixel00[k] = cam.pixel00[k] - 0.5 * (cam.delta_u[k] + cam.delta_v[k]) + 0.5 * f * (cam.delta_u[k] + cam.delta_v[k])
        cam.delta_u[k] *= f
        cam.delta_v[k] *= f
    p.width, p.height, p.samples = w, h, samples
    return scene, cam, p


def test_threads_and_partition_invariance():
    scene, cam, p = small_view(R.SCENE_C2)
    a, sa = O.render(cam, scene, p, threads=1)
    b, sb = O.render(cam, scene, p, threads=8)
    assert np.array_equal(a, b) and sa.segments == sb.segments and sa.camera_rays == 64 * 36 * 4
    parts = []
    for i in range(3):
        p.row_block, p.part_index, p.part_count = 8, i, 3
        img, st = O.render(cam, scene, p, threads=4)
        rows = [r for r in range(36) if (r // 8) % 3 == i]
        assert st.rows == len(rows)
        parts.append((rows, img))
    full = np.empty_like(a)
    for rows, img in parts:
        full[rows] = img
    assert np.array_equal(full, a)


def test_recursion_order_vs_front_to_back():
    """The GPU multiplies col_mod front-to-back; the reference multiplies on the way back up the
    recursion.  Same paths, same draws; colours agree to a few ulp."""
    for which in (R.SCENE_C2, R.SCENE_METAL_TEST):
        scene, cam, p = small_view(which, samples=8)
        a, sa = O.render(cam, scene, p)
        p.flags = R.FLAG_RECURSIVE_ORDER
        b, sb = O.render(cam, scene, p)
        assert sa.segments == sb.segments
        np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("sampler,samples,rays", [(R.SAMPLER_ROW, 10, 10), (R.SAMPLER_STRATIFIED, 10, 16),
                                                  (R.SAMPLER_STRATIFIED, 500, 529), (R.SAMPLER_STRATIFIED, 1000, 1024),
                                                  (R.SAMPLER_CENTRES, 500, 484), (R.SAMPLER_NO_RAND, 77, 1)])
def test_sampler_counts(sampler, samples, rays):
    scene, cam, p = small_view(R.SCENE_C1, 8, 4, samples)
    p.sampler = sampler
    p.depth = 1
    _, st = O.render(cam, scene, p, threads=2)
    assert st.camera_rays == 8 * 4 * rays


def test_bg_color_emission_and_nan_poison():
    light = R.Sphere.new((0, 3, -1), 1.0, (1, 1, 1), R.SCATTER_M)
    for k in range(3):
        light.pod.emitted[k] = 4.0
    scene = R.Scene([R.Sphere.with_albedo((0, -100.5, -1), 100.0, (0.5, 0.5, 0.5)), R.Sphere.with_albedo((0, 0, -1), 0.5, (0.8, 0.3, 0.3)), light],
                    background=(0.05, 0.05, 0.1))
    cam, _ = O.viewport_new(48, np.float32(48) / np.float32(27))
    p = flag_params(depth=6)
    p.width, p.height, p.samples, p.integrator, p.sampler, p.maxt = 48, 27, 8, R.INTEGRATOR_BG_COLOR, R.SAMPLER_ROW, 10000.0
    a, sa = O.render(cam, scene, p)
    p.flags = R.FLAG_RECURSIVE_ORDER
    b, sb = O.render(cam, scene, p)
    assert sa.segments == sb.segments and sa.nan_pixels == sb.nan_pixels
    ok = ~np.isnan(a)
    assert (np.isnan(a) == np.isnan(b)).all() and ok.mean() > 0.9
    np.testing.assert_allclose(a[ok], b[ok], rtol=3e-6, atol=1e-7)
    assert a[ok].max() > 2 * math.sqrt(0.1)      # emission reaches the image: brighter than the gamma-2 background


# ---- 8. statistical tier vs the reference's own objects (lambert + mirror: identical distributions) ---
@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (only exists where /root/reference does)")
def test_oracle_vs_reference_objects_statistical():
    scene = R.Scene.generate(R.SCENE_C1)
    cam, h = O.viewport_new(96, np.float32(96) / np.float32(54))
    p = flag_params(depth=10)
    p.width, p.height, p.samples, p.integrator, p.sampler, p.gamma = 96, 54, 128, R.INTEGRATOR_GRADIENT, R.SAMPLER_ROW, 1.0
    a, sa = O.render(cam, scene, p)
    b, seg, _ = O.ref_render(cam, scene, p, rand_seed=7)
    # 6x6 block means; per-block noise estimated from the within-block spread of both images
    def blocks(x):
        return x[:54, :96].reshape(9, 6, 16, 6, 3).mean(axis=(1, 3))
    da = blocks(a.astype(np.float64)) - blocks(b)
    assert abs(da.mean()) < 2e-3, da.mean()
    assert np.abs(da).max() < 0.06, np.abs(da).max()
    # mean path length agrees (the C++ loop counts one query per ray_colorD call that reaches the loop)
    assert abs(sa.segments / sa.camera_rays - seg / sa.camera_rays) < 0.02


# ---- 9. Rust2 trait surface ----------------------------------------------------------
def rust2_view(w=64, h=36, samples=9, depth=6):
    spheres = [R.Sphere.with_albedo((0, -100.5, -1), 100.0, (0.5, 0.5, 0.5), R.SCATTER_M),       # Lambertian
               R.Sphere.with_albedo((0, 0, -1.2), 0.5, (0.7, 0.3, 0.3), R.SCATTER_M),
               R.Sphere.with_albedo((1.05, 0, -1.2), 0.5, (0.8, 0.6, 0.2), R.METALLIC_M),        # Mirror
               R.Sphere.with_albedo((-1.05, 0, -1.2), 0.5, (1, 1, 1), R.GLASS_M)]                # MirrorGlass{1.5}
    spheres[1].pod.emitted[0] = 0.25                                                              # ColorResult.emmited
    scene = R.Scene(spheres, background=(0.6, 0.7, 0.9))
    cam = R.camera2_new(np.float32(w) / np.float32(h), (0, 0, 0.5), (0, 1, 0), (0, 0, -1), 80.0, 0.01)
    p = flag_params(depth=depth)
    p.width, p.height, p.samples, p.integrator, p.sampler, p.gamma, p.maxt = w, h, samples, R.INTEGRATOR_RUST2, R.SAMPLER_CENTRES, 2.0, 1000.0
    return scene, cam, p


def test_rust2_camera_and_sampler():
    cam = R.camera2_new(2.0, (1, 2, 3), (0, 1, 0), (0, 0, -1), 90.0, 0.1)
    np.testing.assert_allclose(np.array(cam.pixel00), [-2, 1, -1], atol=1e-6)          # left_top = -w - vu/2 - vv/2
    np.testing.assert_allclose(np.array(cam.delta_u), [4, 0, 0], atol=1e-6)            # FULL viewport vectors
    np.testing.assert_allclose(np.array(cam.delta_v), [0, -2, 0], atol=1e-6)
    scene, cam, p = rust2_view(samples=500)
    _, st = O.render(cam, scene, p, threads=4)
    assert st.camera_rays == 64 * 36 * 484                                             # floor(sqrt(500))^2
    assert np.array_equal(R.quantize_u8_rust2(np.float32([0.0, 0.5, 1.0, 2.0])), np.uint8([0, 128, 255, 255]))


def test_rust2_ray_color_semantics():
    scene, cam, p = rust2_view()
    # depth 0 returns the background, not black (Rust2/src/viewport/ray_color.rs:13-16)
    p.depth, p.gamma = 0, 1.0
    img, st = O.render(cam, scene, p, threads=2)
    assert st.segments == 0
    np.testing.assert_allclose(img, np.broadcast_to(np.float32([0.6, 0.7, 0.9]), img.shape), rtol=1e-6)
    # Mirror reflects the UN-normalised direction: |next| == |dir|
    pp = flag_params(depth=1)
    pp.integrator = R.INTEGRATOR_RUST2
    b, _ = O.trace_ray((0, 0, 0.5), (3.0, 0.0, -4.8), 0.0, scene, pp)
    # Lambertian direction is normalised
    lam, _ = O.trace_ray((0, 0, 0.5), (0.0, 0.0, -1.0), 0.0, scene, pp)
    assert lam[0].hit == 1 and lam[0].sphere == 1
    # recursion order vs front-to-back: same paths, colours within a few ulp
    scene, cam, p = rust2_view(samples=16)
    a, sa = O.render(cam, scene, p)
    p.flags = R.FLAG_RECURSIVE_ORDER
    c, sc = O.render(cam, scene, p)
    assert sa.segments == sc.segments
    np.testing.assert_allclose(a, c, rtol=3e-6, atol=1e-7)
    assert a[..., 0].max() > a[..., 2].min()      # the emissive red sphere shows


# ---- 10. the C++ twin's dialect: RTW_FLAG_CPP_DIELECTRIC | RTW_FLAG_CPP_DIFFUSE ------------------
@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (only exists where /root/reference does)")
def test_oracle_cpp_dialect_vs_reference_objects_statistical():
    """metal_test's four spheres (C++/src/tests.cpp:195-212: lambert, mirror, FUZZY3 = 0.7) rendered by the reference's own
    C++ objects and by the oracle with the C++ dialect flags: the fuzzy sphere is where the dialects differ (C++ normalises the
    diffuse and the mirror direction before the lerp, C++/src/materials.cpp:4-13; Rust does not, materials.rs:141-149)."""
    scene = R.Scene.generate(R.SCENE_METAL_TEST)
    cam, h = O.viewport_new(96, np.float32(96) / np.float32(54))
    p = flag_params(depth=10, flags=R.FLAG_CPP)
    p.width, p.height, p.samples, p.integrator, p.sampler, p.gamma = 96, 54, 192, R.INTEGRATOR_GRADIENT, R.SAMPLER_ROW, 1.0
    a, sa = O.render(cam, scene, p)
    b, seg, _ = O.ref_render(cam, scene, p, rand_seed=11)

    def blocks(x):
        return x[:54, :96].reshape(9, 6, 16, 6, 3).mean(axis=(1, 3))
    da = blocks(a.astype(np.float64)) - blocks(b)
    assert abs(da.mean()) < 2e-3, da.mean()
    assert np.abs(da).max() < 0.05, np.abs(da).max()
    assert abs(sa.segments / sa.camera_rays - seg / sa.camera_rays) < 0.02
    # the flags do change the paths (same seed, different image), and only through the documented branches:
    p.flags = 0
    c, _ = O.render(cam, scene, p)
    assert not np.array_equal(a, c)
    p.flags = R.FLAG_CPP_DIELECTRIC                       # no dielectric in this scene: that flag alone changes nothing
    d, _ = O.render(cam, scene, p)
    assert np.array_equal(c, d)


def test_cpp_diffuse_known_answers():
    """One lambert bounce under RTW_FLAG_CPP_DIFFUSE: the scattered direction is a UNIT vector (C++ normalises
    (point + normal + rand_unit) - point), under the Rust rule it is normal + rand_unit (length in [0, 2])."""
    scene = R.Scene([R.Sphere.with_albedo((0, 0, -1), 0.5, (0.5, 0.5, 0.5), R.SCATTER_M)])
    p = flag_params(depth=1)
    p.integrator = R.INTEGRATOR_GRADIENT
    lens = {0: [], R.FLAG_CPP_DIFFUSE: []}
    for flags in lens:
        p.flags = flags
        for s in range(64):
            b, _ = O.trace_ray((0, 0, 0), (0.05, 0.02, -1.0), 0.0, scene, p, pixel=3, sample=s)
            assert b[0].hit == 1
            lens[flags].append(float(np.linalg.norm(np.array(b[0].next_dir, np.float64))))
    assert np.allclose(lens[R.FLAG_CPP_DIFFUSE], 1.0, atol=3e-7)
    assert np.std(lens[0]) > 0.1 and max(lens[0]) <= 2.0 + 1e-6
    # a mirror: Rust reflects unit(d) (already unit), C++ normalises once more -- both unit, both the mirror direction
    scene = R.Scene([R.Sphere.with_albedo((0, 0, -1), 0.5, (1, 1, 1), R.METALLIC_M)])
    dirs = []
    for flags in (0, R.FLAG_CPP_DIFFUSE):
        p.flags = flags
        b, _ = O.trace_ray((0, 0, 0), (0.05, 0.02, -1.0), 0.0, scene, p)
        dirs.append(np.array(b[0].next_dir, np.float64))
    assert np.abs(dirs[0] - dirs[1]).max() < 3e-7 and abs(np.linalg.norm(dirs[1]) - 1.0) < 3e-7


def test_chunk_sums_flag_on_the_oracle():
    """RTW_FLAG_CHUNK_SUMS (rtw.h) on the CPU side: <= 4 samples is the plain left-to-right sum; more samples differ from it by
    rounding only, and threads / partitions still do not matter."""
    scene, cam, p = small_view(R.SCENE_C2, 48, 27, 4)
    p.gamma = 1.0
    a, _ = O.render(cam, scene, p)
    p.flags = R.FLAG_CHUNK_SUMS
    b, _ = O.render(cam, scene, p)
    assert np.array_equal(a, b)
    p.samples, p.flags = 23, 0
    a, _ = O.render(cam, scene, p, threads=1)
    p.flags = R.FLAG_CHUNK_SUMS
    b, _ = O.render(cam, scene, p, threads=1)
    c, _ = O.render(cam, scene, p, threads=8)
    assert np.array_equal(b, c) and not np.array_equal(a, b)
    assert (np.abs(a - b) / np.maximum(np.abs(a), 1e-6)).max() < 1e-5
