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
    assert L.rtw_abi_version() == 4


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
    assert C.sizeof(R.RtwStats) == 160
    assert C.sizeof(R.RtwScene) == 96
    assert C.sizeof(R.RtwQuad) == 88
    assert C.sizeof(R.RtwInstance) == 48


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


# ---- scene wire format + image writers -----------------------------------------
def test_json_round_trip_like_deserialize_test():
    """Rust/src/viewport/json_tests.rs:48-92: scene -> JsonValue -> Scene compares equal."""
    import json
    scene = R.Scene.new_sphere([R.Sphere.new((-0.5, 0.0, -1.0), 0.5, (0.6, 0.6, 0.6), R.SCATTER_M),
                                R.Sphere.new((0.5, 0.0, -1.0), 0.5, (1.0, 1.0, 1.0), R.SCATTER_M),
                                R.Sphere.new((0.0, 0.0, -2.0), 1.0, (0.5, 1.0, 0.0), R.METALLIC_M),
                                R.Sphere.new_moving((2.4, 0.0, -0.8), 1.4, (0.9, 0.9, 0.9), R.GLASS_M, (0.0, 60.0, 0.0))])
    text = scene.to_json()
    doc = json.loads(text)                                         # it is valid JSON with the reference's members
    assert set(doc["spheres"][0]) == {"origin", "radius", "col_mod", "material", "velocity", "texture"}
    assert set(doc["spheres"][0]["material"]) == {"metallicness", "opacity", "ir"}
    assert doc["spheres"][3]["velocity"] == {"x": 0, "y": 60, "z": 0}
    assert doc["spheres"][0]["texture"] == {"row": 1, "col": 1, "img": [doc["spheres"][0]["col_mod"]]}
    back = R.Scene.from_json(text)
    assert back.n_spheres == 4
    assert bytes(back._spheres)[: 80 * 4] == bytes(scene._spheres)[: 80 * 4]      # bit-exact f32 round trip


def test_json_textured_scene_and_cpp_dialect_and_errors():
    sc = R.Scene.generate(R.SCENE_C5)
    back = R.Scene.from_json(sc.to_json())
    assert back.n_spheres == sc.n_spheres and back.n_textures == 1 and back.n_texels == 8
    assert bytes(back._spheres)[: 80 * sc.n_spheres] == bytes(sc._spheres)[: 80 * sc.n_spheres]
    assert np.array_equal(back._texels[:8], sc._texels[:8])
    # the C++ dialect has neither velocity nor texture (C++/headers/sphere.h:28-31)
    cpp = '{"spheres":[{"origin":{"x":-0.52,"y":0,"z":-1.2},"radius":0.4,"material":{"metallicness":1,"opacity":0,"ir":1},"col_mod":{"x":0.7,"y":0.7,"z":0.7}}]}'
    s = R.Scene.from_json(cpp)
    assert s.n_spheres == 1 and s._spheres[0].tex == -1 and list(s._spheres[0].tex_color) == [1, 1, 1]
    assert abs(s._spheres[0].center[0] + 0.52) < 1e-7
    for bad in ('{"spheres":3}', '{"spheres":[{"origin":{"x":0,"y":0},"radius":1}]}', '{"spheres":[', 'nonsense', '{}'):
        with pytest.raises(R.RtwError):
            R.Scene.from_json(bad)


def test_png_and_ppm_writers(tmp_path):
    import struct, zlib
    rng = np.random.default_rng(1)
    img = rng.uniform(-0.1, 1.1, (7, 5, 3)).astype(np.float32)
    png = str(tmp_path / "a.png")
    R.write_img_f32(img, png)
    raw = open(png, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, {}
    while pos < len(raw):
        n, t = struct.unpack(">I4s", raw[pos:pos + 8])
        data = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(t + data) & 0xFFFFFFFF
        chunks[t] = data
        pos += 12 + n
    w, h, depth, ctype = struct.unpack(">IIBB", chunks[b"IHDR"][:10])
    assert (w, h, depth, ctype) == (5, 7, 8, 2)
    pix = np.frombuffer(zlib.decompress(chunks[b"IDAT"]), np.uint8).reshape(7, 1 + 5 * 3)
    assert (pix[:, 0] == 0).all() and np.array_equal(pix[:, 1:].reshape(7, 5, 3), R.quantize_u8(img))
    ppm = str(tmp_path / "a.ppm")
    R.write_ppm(ppm, np.clip(img, 0, 1))
    tok = open(ppm).read().split()
    assert tok[:4] == ["P3", "5", "7", "255"]
    want = (255 * np.clip(img, 0, 1).astype(np.float64)).astype(np.int32)       # RGB.cpp:16-20 truncation
    assert np.array_equal(np.array(tok[4:], np.int32).reshape(7, 5, 3), want)


def test_missing_hip_library_fails_loudly():
    """The product has no CPU fallback: without librtw_hip.so the package refuses to work (ImportError)."""
    import subprocess, sys
    code = ("import os, sys; os.environ['RTW_HIP_LIB'] = '/nonexistent/librtw_hip.so'; sys.path.insert(0, %r);\n"
            "import rtw_amd as R\n"
            "try:\n    R.lib(); print('LOADED')\nexcept ImportError as e:\n    print('IMPORTERROR', 'no CPU fallback' in str(e))\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert "IMPORTERROR True" in out.stdout, out.stdout + out.stderr


def _validate(scene, t0=0.0, t1=0.0):
    n, d, b, h = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc = R.lib().rtw_bvh_validate(C.byref(scene.pod), t0, t1, C.byref(n), C.byref(d), C.byref(b), C.byref(h))
    return rc, n.value, d.value, b.value, h.value


def test_bvh_builder_invariants():
    """The host BVH (csrc/rtw_host.cpp build_bvh): every sphere exactly once, nested time-expanded bounds, f16 copy
    contains the f32 boxes, depth within the device stack -- for the config scenes and for adversarial ones."""
    rc, n, d, b, h = _validate(R.Scene.generate(R.SCENE_C2))
    assert (rc, n, b, h) == (0, 483, 1, 1) and d <= 32            # ground kept outside the tree, 484 leaves
    rc, n, d, b, h = _validate(R.Scene.generate(R.SCENE_C5), 0.0, 1 / 30)
    assert (rc, b, h) == (0, 1, 1)
    rc, n, d, b, h = _validate(R.Scene.generate(R.SCENE_C4))
    assert (rc, n, b) == (0, 181, 1)
    rng = np.random.default_rng(11)
    # many coincident centres, a line of spheres, wildly different radii, big coordinates (no f16 copy), > 512 nodes
    cases = {
        "coincident": [R.Sphere.new((0, 0, 0), 0.1 + 0.001 * i) for i in range(70)],
        "line": [R.Sphere.new((i * 0.5, 0, 0), 0.3) for i in range(200)],
        "radii": [R.Sphere.new(rng.uniform(-5, 5, 3), float(10 ** rng.uniform(-3, 2))) for _ in range(120)],
        "far": [R.Sphere.new(rng.uniform(-1, 1, 3) + 5e4, 0.5) for _ in range(40)],
        "many": [R.Sphere.new(rng.uniform(-30, 30, 3), 0.4) for _ in range(900)],
        "moving": [R.Sphere.new_moving(rng.uniform(-5, 5, 3), 0.4, None, None, rng.uniform(-20, 20, 3)) for _ in range(100)],
        "one": [R.Sphere.new((0, 0, -1), 0.5)], "two": [R.Sphere.new((0, 0, -1), 0.5), R.Sphere.new((1, 0, -1), 0.5)], "none": [],
    }
    for name, spheres in cases.items():
        t1 = 0.5 if name == "moving" else 0.0
        rc, n, d, b, h = _validate(R.Scene(spheres), 0.0, t1)
        assert rc == 0 and d <= 32, name
        if name == "far":
            assert h == 0                       # coordinates beyond the f16 guard: global f32 nodes only
        if name == "many":
            assert h == 0 and n + b + 1 == 900  # more than 512 nodes: no LDS copy
        if name in ("one", "none"):
            assert n == 0


# ---- quads / instances: host constructors and generators (no GPU) ---------------------------------------
def test_quad_and_box_constructors():
    q = R.Quad.new((1, 2, 3), (4, 0, 0), (0, 5, 0), R.METALLIC_M, (0.1, 0.2, 0.3), emitted=(7, 8, 9)).pod
    assert list(q.origin) == [1, 2, 3] and list(q.u) == [4, 0, 0] and list(q.v) == [0, 5, 0]
    assert (q.metallicness, q.opacity, q.ir, q.tex) == (1.0, 0.0, 1.0, -1)
    assert np.allclose(list(q.tex_color), [0.1, 0.2, 0.3]) and list(q.emitted) == [7, 8, 9] and list(q.velocity) == [0, 0, 0]
    # Instance::new_box (instance.rs:83-176): front, right, back, left, top, bottom -- origins and edge vectors as written there
    box = R.Instance.new_box((1.0, 0.5, 0.5), (-1.0, -0.5, -0.5), (0.2, 0.2, 0.2), R.SCATTER_M)     # corners in any order (minf/maxf)
    want = [((-1, -.5, .5), (2, 0, 0), (0, 1, 0)), ((1, -.5, .5), (0, 0, -1), (0, 1, 0)), ((1, -.5, -.5), (-2, 0, 0), (0, 1, 0)),
            ((-1, -.5, -.5), (0, 0, 1), (0, 1, 0)), ((-1, .5, .5), (2, 0, 0), (0, 0, -1)), ((-1, -.5, -.5), (2, 0, 0), (0, 0, 1))]
    for quad, (o, u, v) in zip(box.quads, want):
        assert np.array_equal(list(quad.origin), o) and np.array_equal(list(quad.u), u) and np.array_equal(list(quad.v), v)
    # translate / rotate accumulate in f32 like `+=` (instance.rs:234-243)
    box.translate((0.1, 0.2, 0.3)); box.translate((0.1, 0.2, 0.3))
    assert box.translation == [float(np.float32(0.1) + np.float32(0.1)), float(np.float32(0.2) + np.float32(0.2)), float(np.float32(0.3) + np.float32(0.3))]


def test_scene_generate_geom_counts_capacity_and_pools():
    L = R.lib()
    counts = (C.c_uint32 * 5)()
    bg = (C.c_float * 3)(9, 9, 9)
    assert L.rtw_scene_generate_geom(R.SCENE_PRESENTATION, 42, None, None, None, None, None, None, counts, bg) == 0
    assert list(counts) == [1, 6, 2, 0, 8] and list(bg) == [0, 0, 0]
    quads = (R.RtwQuad * 6)()
    small = (C.c_uint32 * 5)(1, 5, 2, 0, 8)                      # one quad short
    assert L.rtw_scene_generate_geom(R.SCENE_PRESENTATION, 42, None, quads, None, None, None, small, counts, bg) == -1
    assert L.rtw_scene_generate_geom(99, 42, None, None, None, None, None, None, counts, bg) == -1
    sc = R.Scene.generate_geom(R.SCENE_PRESENTATION)
    smoke, glass = sc._instances[0], sc._instances[1]
    assert (smoke.first_quad, smoke.n_quads, smoke.medium, smoke.density) == (0, 6, R.MEDIUM_CONST_DENSITY, 2.0)
    assert (glass.first_quad, glass.n_quads, glass.medium) == (6, 2, R.MEDIUM_SURFACE)
    assert abs(smoke.rotation[1] - np.pi / 4) < 1e-7 and abs(glass.rotation[1] + np.pi / 6) < 1e-7
    assert abs(glass.translation[0] - (2.0 - np.sqrt(np.float32(3.0)))) < 1e-7
    assert [q.ir for q in list(sc._inst_quads)[6:8]] == [1.5, float(np.float32(2.0) / np.float32(3.0))]
    light = sc._quads[1]
    assert list(light.emitted) == [4, 4, 4] and (light.metallicness, light.opacity, light.ir) == (0.0, 0.0, 1.0)
    # the sphere-only ids also work through the same entry point
    assert L.rtw_scene_generate_geom(R.SCENE_C1, 42, None, None, None, None, None, None, counts, bg) == 0 and list(counts) == [3, 0, 0, 0, 0]
    # the JSON wire format stays the reference's: spheres only (viewport.rs:174-180)
    assert '"spheres"' in sc.to_json() and "quads" not in sc.to_json()


def test_python_scene_flattens_instances_into_pools():
    a = R.Instance.new([R.Sphere.new((0, 0, 0), 1.0)], [R.Quad.new((0, 0, 0), (1, 0, 0), (0, 1, 0))])
    b = R.Instance.new_quads([R.Quad.new((0, 0, 1), (1, 0, 0), (0, 1, 0)), R.Quad.new((0, 0, 2), (1, 0, 0), (0, 1, 0))])
    sc = R.Scene.new([], [R.Quad.new((0, 0, 3), (1, 0, 0), (0, 1, 0))], [a, b])
    assert (sc.pod.n_spheres, sc.pod.n_quads, sc.pod.n_instances, sc.pod.n_inst_spheres, sc.pod.n_inst_quads) == (0, 1, 2, 1, 3)
    i0, i1 = sc._instances[0], sc._instances[1]
    assert (i0.first_sphere, i0.n_spheres, i0.first_quad, i0.n_quads) == (0, 1, 0, 1)
    assert (i1.first_sphere, i1.n_spheres, i1.first_quad, i1.n_quads) == (1, 0, 1, 2)
    assert [q.origin[2] for q in list(sc._inst_quads)[:3]] == [0.0, 1.0, 2.0]


def test_quantiser_against_the_reference_writer_test_image():
    """Rust/test.png is what the reference's own writer test saves (write_img.rs:33-58): pixel (i, j) = (round(255 * i/255),
    round(255 * j/255), round(255 * 0.25)) with f32 arithmetic.  rtw_quantize_u8 (== write_img_f32's rule, write_img.rs:11-15) and
    the PNG writer must reproduce it exactly."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_images.npz"))
    gold = np.empty((256, 256, 3), np.uint8)            # the fixture stores the image's three separable profiles
    gold[:, :, 0] = z["write_test_r_of_column"][None, :]
    gold[:, :, 1] = z["write_test_g_of_row"][:, None]
    gold[:, :, 2] = z["write_test_b"][0]
    i = np.arange(256, dtype=np.float32) / np.float32(255)
    img = np.empty((256, 256, 3), np.float32)
    img[:, :, 0] = i[None, :]
    img[:, :, 1] = i[:, None]
    img[:, :, 2] = np.float32(0.25)
    assert np.array_equal(R.quantize_u8(img), gold)


# ---- round 2: multi-GPU entry points, builder depth bound, bounded JSON reads, the reference's rotation KATs ----------
def test_multi_gpu_entry_points_without_a_device():
    """rtw_mgpu_* / rtw_render_multi_gpu (the fork / ordered join of viewport.rs:236-244 over GPUs) fail loudly without
    a GPU: RTW_E_NO_DEVICE, never a CPU render."""
    if R.device_count() > 0:
        pytest.skip("a GPU is present")
    L = R.lib()
    h = C.c_void_p()
    dev = (C.c_int * 3)(0, 0, 0)
    assert L.rtw_mgpu_create(dev, 3, C.byref(h)) == -2 and not h.value
    assert L.rtw_mgpu_create(None, 3, C.byref(h)) == -1 and L.rtw_mgpu_create(dev, 0, C.byref(h)) == -1
    scene = R.Scene.generate(R.SCENE_C1)
    cam, p = R.default_view(R.SCENE_C1)
    out = np.zeros((p.height, p.width, 3), np.float32)
    assert L.rtw_render_multi_gpu(dev, 3, C.byref(cam), C.byref(scene.pod), C.byref(p), out.ctypes.data_as(C.c_void_p), None) == -2
    assert not out.any()
    with pytest.raises(R.RtwError):
        R.MultiRenderer([0, 0])
    L.rtw_mgpu_destroy(None)            # like free(NULL)
    assert L.rtw_mgpu_render(None, C.byref(cam), C.byref(p), out.ctypes.data_as(C.c_void_p), None, None) == -1


def geometric_scene(n=400, ratio=1.2, rel_radius=None):
    """Spheres at x = ratio^i: every SAH split peels a few spheres off the dense end (equal spheres, n = 400,
    ratio 1.2 built a tree of depth 27).  rel_radius: radius = rel_radius * x (visible from a camera near the dense end)."""
    f = np.float32
    mats = [R.SCATTER_M, R.METALLIC_M, R.GLASS_M]
    return R.Scene([R.Sphere.with_albedo((float(f(ratio) ** i), 0.0, -5.0), 0.01 if rel_radius is None else float(f(rel_radius) * f(ratio) ** i),
                                         (0.5, 0.5, 0.5), mats[i % 3]) for i in range(n)])


@pytest.mark.parametrize("n,ratio", [(400, 1.2), (3000, 1.01), (64, 4.0), (1000, 1.05)])
def test_bvh_depth_is_bounded_by_construction(n, ratio):
    """The builder's invariant depth + ceil(log2(count)) <= RTW_BVH_STACK (24) holds on skewed scenes."""
    sc = geometric_scene(n, ratio)
    nn, depth, nbig, f16 = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    assert R.lib().rtw_bvh_validate(C.byref(sc.pod), 0.0, 0.0, C.byref(nn), C.byref(depth), C.byref(nbig), C.byref(f16)) == 0
    assert depth.value <= 24 and nn.value + nbig.value + 1 >= n


def test_json_parser_never_reads_past_len():
    """rtw_scene_from_json takes (text, len): literals and numbers at the very end of a buffer that is NOT
    NUL-terminated must not be scanned past `len`."""
    L = R.lib()
    ns = C.c_uint32()
    doc = b'{"spheres":[]}'
    # the document followed by bytes that would extend a number / complete a literal if the parser ran on
    for tail in (b"123456", b"ue", b"e+9", b"]}"):
        buf = C.create_string_buffer(doc + tail, len(doc) + len(tail))       # no trailing NUL inside the window
        assert L.rtw_scene_from_json(buf, len(doc), None, 0, C.byref(ns), None, 0, None, None, 0, None) == 0 and ns.value == 0
    for frag in (b"tr", b"12", b"-", b'{"spheres":[1.5', b'{"spheres":[tru', b"nul"):
        buf = C.create_string_buffer(frag + b"e9999}]}", len(frag) + 8)
        assert L.rtw_scene_from_json(buf, len(frag), None, 0, C.byref(ns), None, 0, None, None, 0, None) == -1
    # texture dimensions are validated before they are converted to integers
    sph = ('{"origin":{"x":0,"y":0,"z":0},"radius":1,"col_mod":{"x":1,"y":1,"z":1},"material":{"metallicness":0,"opacity":0,"ir":1},'
           '"texture":{"row":%s,"col":%s,"img":[{"x":1,"y":1,"z":1}]}}')
    for row, col in (("-1", "1"), ("1e30", "1"), ("1", "-5"), ("0", "1"), ("4294967296", "1")):
        raw = ('{"spheres":[' + sph % (row, col) + "]}").encode()
        assert L.rtw_scene_from_json(raw, len(raw), None, 0, C.byref(ns), None, 0, None, None, 0, None) == -1, (row, col)
    raw = ('{"spheres":[' + sph % ("1", "1") + "]}").encode()
    assert L.rtw_scene_from_json(raw, len(raw), None, 0, C.byref(ns), None, 0, None, None, 0, None) == 0 and ns.value == 1


def test_rotated_known_answers_of_the_reference():
    """The reference's own rotation_tests (Rust/src/vec3.rs:363-404; its Vec3 == is |d| < 1e-7 per component,
    vec3.rs:17-21) on the host mirror and on the oracle, which must also agree with each other bit for bit."""
    from tests import oracle_binding as O
    PI = np.float32(np.pi)
    rot1 = (float(PI / np.float32(6.0)), 0.0, 0.0)
    rot2 = (0.0, 0.0, float(PI / np.float32(6.0)))
    c, s = np.float32(np.cos(np.float32(rot2[2]))), np.float32(np.sin(np.float32(rot2[2])))
    cases = [((1.0, 0.0, 0.0), rot1, (1.0, 0.0, -0.0)),                   # x: rot1.y.cos() * rot1.z.cos() == 1
             ((1.0, 0.0, 0.0), rot2, (c, s, 0.0)),
             ((1.0, 0.0, 1.0), rot2, (c, s, 1.0))]
    for v, rot, want in cases:
        a, b = R.vec3_rotated(v, rot), O.rotated(v, rot)
        assert a.tobytes() == b.tobytes()
        assert np.abs(a - np.float32(want)).max() < 1e-7, (v, rot, a, want)


def test_tile_order_modes_on_the_bench_frame():
    """RTW_OPT_TILE_ORDER on the headline frame (1920x1080 = 240 x 135 tiles): 0 raster, 3 reverse raster, 1 groups of 8 consecutive
    tiles scattered (neighbouring queue positions inside a group stay neighbours on screen, groups jump), 2 expensive tiles first by
    the centre-ray estimate: the sphere field nearest the camera (bottom of this image) first, sky last."""
    scene = R.Scene.generate(R.SCENE_C2)
    cam, p = R.default_view(R.SCENE_C5)
    n = 240 * 135
    buf = (C.c_uint32 * n)()
    order = {}
    for mode in (0, 1, 2, 3, 4):
        assert R.lib().rtw_tile_order(mode, 1920, 1080, C.byref(cam), C.byref(scene.pod), buf, n) == 0
        order[mode] = np.array(buf[:])
        assert np.array_equal(np.sort(order[mode]), np.arange(n))
    assert np.array_equal(order[0], np.arange(n)) and np.array_equal(order[3], np.arange(n)[::-1])
    g = order[1].reshape(-1, 8)
    assert (np.diff(g, axis=1) == 1).all() and (g[:, 0] % 8 == 0).all()                  # groups of 8 consecutive tiles
    assert np.abs(np.diff(g[:, 0])).min() > 8 * 100                                        # consecutive groups are far apart
    rows = order[2] // 240
    assert rows[:2000].mean() > 100 and rows[-2000:].mean() < 10                            # near field first, sky (top rows) last
    rows4 = order[4] // 240                                                                  # mode 4: the same, in coarse steps with raster runs inside
    assert rows4[:2000].mean() > 100 and rows4[-2000:].mean() < 10
    assert (np.diff(order[4]) > 0).mean() > 0.99 and (np.diff(order[2]) > 0).mean() < 0.9
    # without a scene there is nothing to estimate from: mode 2 keeps the scattered order of mode 1
    assert R.lib().rtw_tile_order(2, 1920, 1080, C.byref(cam), None, buf, n) == 0 and np.array_equal(np.array(buf[:]), order[1])
    # a camera that looks UP from below the ground plane region: the estimate follows the camera, not the image rows
    vp = R.Viewport.new_from_res(1920, 1080, 1, 1, 1.0, vfov=20.0, origin=(13.0, 2.0, 3.0), direction=(-0.9636, -0.1482, -0.2224), vup=(0.0, -1.0, 0.0))
    cam2 = vp.camera()
    assert R.lib().rtw_tile_order(2, 1920, 1080, C.byref(cam2), C.byref(scene.pod), buf, n) == 0
    rows2 = np.array(buf[:]) // 240
    assert rows2[:2000].mean() < 35 and rows2[-2000:].mean() > 125                           # upside-down image: the near field is at the top now


def test_one_hip_runtime_in_this_process():
    """conftest imports torch before the library is loaded, so both share ONE copy of libamdhip64 (rtw_ctx_create refuses a process with two)."""
    assert R.lib().rtw_hip_runtime_count() == 1
    assert "libamdhip64" in R.lib().rtw_strerror(-8).decode() and "torch" in R.lib().rtw_strerror(-8).decode()
