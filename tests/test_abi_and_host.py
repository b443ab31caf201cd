"""CPU-side checks of the product library: it loads, exports every symbol include/rtw.h declares,
and its host logic (constructors, partitioning, quantisation, scene generators, argument checking)
behaves like the reference's.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import rtw_amd as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rtw_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = R.lib()
    names = declared_functions("rtw.h")
    assert len(names) >= 17, names
    for n in names:
        assert hasattr(L, n), f"librtw_hip.so does not export {n}"
    assert L.rtw_abi_version() == 1


def test_oracle_exports_every_declared_symbol():
    from tests import oracle_binding as O
    for n in declared_functions("rtw_oracle.h"):
        assert hasattr(O.lib(), n), n


def test_pod_sizes_match_header():
    # the C structs are naturally packed 4-byte fields (+ one u64): sizes are part of the ABI
    assert C.sizeof(R.RtwCamera) == 84
    assert C.sizeof(R.RtwSphere) == 80
    assert C.sizeof(R.RtwTexture) == 16
    assert C.sizeof(R.RtwParams) == 72
    assert C.sizeof(R.RtwStats) == 96
    assert C.sizeof(R.RtwScene) == 48


def test_no_device_is_an_error_not_a_fallback():
    """Without a GPU the render entry points must fail loudly."""
    if R.device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert R.lib().rtw_ctx_create(0, C.byref(h)) == -2          # RTW_E_NO_DEVICE
    scene = R.Scene.generate(R.SCENE_C1)
    cam, p = R.default_view(R.SCENE_C1)
    out = np.zeros((p.height, p.width, 3), np.float32)
    assert R.lib().rtw_render(C.byref(cam), C.byref(scene.pod), C.byref(p), out.ctypes.data_as(C.c_void_p), None) == -2
    with pytest.raises(R.RtwError):
        R.Renderer(0)


def test_strerror_and_invalid_arguments():
    L = R.lib()
    assert L.rtw_strerror(0) == b"ok" and b"invalid" in L.rtw_strerror(-1)
    cam, h = R.RtwCamera(), C.c_uint32()
    assert L.rtw_viewport_new(0, 1.0, None, None, None, None, None, C.byref(cam), C.byref(h)) == -1
    assert L.rtw_viewport_new_from_res(10, 0, None, None, None, None, None, C.byref(cam), C.byref(h)) == -1
    assert L.rtw_sphere_new(None, 1.0, None, None, None, None) == -1
    assert L.rtw_scene_generate(99, 1, None, 0, None, None, 0, None, None, 0, None) == -1


def test_part_rows():
    L = R.lib()
    assert L.rtw_part_rows(1080, 8, 0, 1) == 1080
    assert sum(L.rtw_part_rows(1080, 8, i, 8) for i in range(8)) == 1080
    assert [L.rtw_part_rows(20, 8, i, 2) for i in range(2)] == [12, 8]
    assert L.rtw_part_rows(20, 0, 0, 2) == 0 and L.rtw_part_rows(20, 8, 2, 2) == 0


def test_quantize_u8_matches_write_img_rule():
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-0.2, 1.3, 5000), [0.0, 1.0, 0.5 / 255, 1.5 / 255, 254.5 / 255, np.nan, np.inf, -np.inf]]).astype(np.float32)
    got = R.quantize_u8(x)
    v = x * np.float32(255.0)
    want = np.where(np.isnan(v), 0, np.floor(np.clip(v, 0, 255).astype(np.float64) + 0.5)).astype(np.uint8)   # round half away from zero, v >= 0
    assert np.array_equal(got, want)


def test_sphere_constructors_mirror_the_reference():
    s = R.Sphere.new((1, 2, 3), 0.5).pod
    assert list(s.col_mod) == [1, 1, 1] and list(s.tex_color) == [1, 1, 1] and s.tex == -1
    assert (s.metallicness, s.opacity, s.ir) == R.EMPTY_M and list(s.velocity) == [0, 0, 0]
    s = R.Sphere.new((0, 0, 0), 1.0, (0.8, 0.5, 1.0), R.GLASS_M).pod
    np.testing.assert_array_equal(np.float32(list(s.col_mod)), np.float32([0.8, 0.5, 1.0]))
    np.testing.assert_array_equal(np.float32(list(s.tex_color)), np.float32([0.8, 0.5, 1.0]))     # the c*c quirk
    assert s.opacity == 1.0 and s.ir == 1.5
    m = R.Sphere.new_moving((0, 0, 0), 1.0, None, None, (0, 60, 0)).pod
    assert list(m.velocity) == [0, 60, 0]
    t = R.Sphere.new_with_texture((0, 0, 0), 1.0, None, None, 3).pod
    assert t.tex == 3 and list(t.col_mod) == [1, 1, 1]
    assert abs(R.GLASSR_M[2] - np.float32(1 / 1.5)) < 1e-7


@pytest.mark.parametrize("which,n", [(R.SCENE_C1, 3), (R.SCENE_METAL_TEST, 4), (R.SCENE_C2, 485), (R.SCENE_C4, 183), (R.SCENE_C5, 485)])
def test_scene_generators_are_deterministic(which, n):
    a, b = R.Scene.generate(which, 42), R.Scene.generate(which, 42)
    assert a.n_spheres == n == b.n_spheres
    assert bytes(a._spheres)[: 80 * n] == bytes(b._spheres)[: 80 * n]
    if which in (R.SCENE_C2, R.SCENE_C5):
        c = R.Scene.generate(which, 43)
        assert bytes(a._spheres)[: 80 * min(n, c.n_spheres)] != bytes(c._spheres)[: 80 * min(n, c.n_spheres)]
    if which == R.SCENE_C5:
        assert a.n_textures == 1 and a.n_texels == 8 and a._spheres[0].tex == 0
        assert any(a._spheres[i].velocity[1] != 0 for i in range(n))
    cam, p = R.default_view(which)
    assert p.width > 0 and p.height > 0 and p.depth in (10, 50)


def test_default_views_match_baseline_configs():
    sizes = {R.SCENE_C1: (400, 225, 10, 10), R.SCENE_C2: (1200, 675, 100, 50), R.SCENE_C4: (1920, 1080, 1000, 50),
             R.SCENE_C5: (1920, 1080, 500, 50)}
    for which, (w, h, spp, depth) in sizes.items():
        cam, p = R.default_view(which)
        assert (p.width, p.height, p.samples, p.depth) == (w, h, spp, depth)
    cam, _ = R.default_view(R.SCENE_C5)
    assert abs(cam.shutter - 1 / 30) < 1e-8 and cam.time0 == 0.0


def test_python_viewport_mirrors_the_pod():
    vp = R.Viewport.new_from_res(400, 225, 10, 10, 2.0)
    assert (vp.width, vp.height) == (400, 225)
    vp.frame, vp.fps, vp.shutter_speed = 3, 60.0, 0.25
    cam = vp.camera()
    assert cam.time0 == np.float32(3) / np.float32(60) and cam.shutter == 0.25
    p = vp.params(R.INTEGRATOR_BG_COLOR, R.SAMPLER_ROW, R.ACCEL_BRUTE)
    assert (p.integrator, p.sampler, p.accel, p.samples, p.depth) == (1, 0, 0, 10, 10)
