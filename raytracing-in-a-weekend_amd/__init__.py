"""rtw-mi355x: MI355X-native path tracer behind the reference's Viewport/Scene/Sphere/Material surface.

This package is a thin ctypes binding over ``librtw_hip.so`` (the product: hand-written HIP kernels
for gfx950 + the C ABI of ``include/rtw.h`` + the C++ host mirror of the reference constructors).
Python only moves pointers; it does no arithmetic on the render path, and there is NO CPU fallback:
if the HIP library is missing or no GPU is present, rendering raises.

Reference surface mirrored (names and argument meaning):
  Viewport.new_from_res / Viewport.new      Rust/src/viewport.rs:308-428
  Viewport.render / async_render            Rust/src/viewport.rs:215-248,430-478
  Scene.new_sphere                          Rust/src/viewport.rs:90-105
  Scene.new / new_quad                      Rust/src/viewport.rs:106-135
  Sphere.new / new_moving / new_with_texture Rust/src/objects/sphere.rs:151-247
  Quad.new                                  Rust/src/objects/quad.rs:84-110
  Instance.new / new_quads / new_sphere / new_box, translate, rotate   Rust/src/objects/instance.rs:83-248
  METALLIC_M, SCATTER_M, FUZZY3_M, GLASS_M, GLASSR_M   Rust/src/objects/materials.rs:157-212

The directory name carries a hyphen (it is fixed by the build contract); import it with
``importlib.import_module("raytracing-in-a-weekend_amd")`` or through the ``rtw_amd`` alias module
at the repo root.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTW_HIP_LIB", os.path.join(_HERE, "librtw_hip.so"))   # override: A/B builds of the same ABI

# ---- enums (include/rtw.h) ---------------------------------------------------------------------
RTW_OK = 0
INTEGRATOR_GRADIENT, INTEGRATOR_BG_COLOR, INTEGRATOR_NORMAL, INTEGRATOR_FLAG, INTEGRATOR_RUST2 = 0, 1, 2, 3, 4
SAMPLER_ROW, SAMPLER_STRATIFIED, SAMPLER_CENTRES, SAMPLER_NO_RAND = 0, 1, 2, 3
ACCEL_BRUTE, ACCEL_BVH = 0, 1
FLAG_RECURSIVE_ORDER, FLAG_CPP_DIELECTRIC, FLAG_GLOBAL_NODES, FLAG_CPP_DIFFUSE, FLAG_CHUNK_SUMS = 1, 2, 4, 8, 16
FLAG_CPP = FLAG_CPP_DIELECTRIC | FLAG_CPP_DIFFUSE      # what Viewport::RenderGPU of the C++ tree asks for
OPT_CHUNK_LEN, OPT_SAMPLE_BANK_GB, OPT_LDS_GEOM, OPT_BLOCKS_PER_CU, OPT_LIST_WALK_MAX, OPT_TILE_ORDER, OPT_GRAB_BLOCKS, OPT_SUB_QUEUES = 1, 2, 3, 4, 5, 6, 7, 8
OPT_TAIL_UNITS = 9
SCENE_C1, SCENE_C2, SCENE_C4, SCENE_C5, SCENE_METAL_TEST, SCENE_QUAD_TEST, SCENE_PRESENTATION, SCENE_FIRST_FRAME = 1, 2, 4, 5, 6, 7, 8, 9
MEDIUM_SURFACE, MEDIUM_CONST_DENSITY = 0, 1

# materials.rs:157-212 presets as (metallicness, opacity, ir)
METALLIC_M = (1.0, 0.0, 1.0)
SCATTER_M = (0.0, 0.0, 1.0)
FUZZY3_M = (0.7, 0.0, 1.0)
GLASS_M = (1.0, 1.0, 1.5)
GLASSR_M = (1.0, 1.0, float(np.float32(1.0) / np.float32(1.5)))
EMPTY_M = SCATTER_M


class RtwError(RuntimeError):
    def __init__(self, status: int, what: str):
        super().__init__(f"{what}: status {status} ({_strerror(status)})")
        self.status = status


# ---- PODs ----------------------------------------------------------------------------------------
class RtwCamera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3),
                ("pixel00", C.c_float * 3), ("delta_u", C.c_float * 3), ("delta_v", C.c_float * 3),
                ("lens_radius", C.c_float), ("time0", C.c_float), ("shutter", C.c_float)]


class RtwSphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("velocity", C.c_float * 3),
                ("col_mod", C.c_float * 3), ("tex_color", C.c_float * 3),
                ("metallicness", C.c_float), ("opacity", C.c_float), ("ir", C.c_float),
                ("emitted", C.c_float * 3), ("tex", C.c_int32)]


class RtwTexture(C.Structure):
    _fields_ = [("row", C.c_uint32), ("col", C.c_uint32), ("texel_offset", C.c_uint32), ("emit_tex", C.c_uint32)]


class RtwQuad(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3), ("velocity", C.c_float * 3),
                ("tex_color", C.c_float * 3), ("metallicness", C.c_float), ("opacity", C.c_float), ("ir", C.c_float),
                ("emitted", C.c_float * 3), ("tex", C.c_int32)]


class RtwInstance(C.Structure):
    _fields_ = [("first_sphere", C.c_uint32), ("n_spheres", C.c_uint32), ("first_quad", C.c_uint32), ("n_quads", C.c_uint32),
                ("translation", C.c_float * 3), ("rotation", C.c_float * 3), ("density", C.c_float), ("medium", C.c_uint32)]


class RtwScene(C.Structure):
    _fields_ = [("spheres", C.POINTER(RtwSphere)), ("textures", C.POINTER(RtwTexture)),
                ("texels", C.POINTER(C.c_float)), ("n_spheres", C.c_uint32), ("n_textures", C.c_uint32),
                ("n_texels", C.c_uint32), ("background", C.c_float * 3),
                ("quads", C.POINTER(RtwQuad)), ("instances", C.POINTER(RtwInstance)),
                ("inst_spheres", C.POINTER(RtwSphere)), ("inst_quads", C.POINTER(RtwQuad)),
                ("n_quads", C.c_uint32), ("n_instances", C.c_uint32), ("n_inst_spheres", C.c_uint32), ("n_inst_quads", C.c_uint32)]


class RtwParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("samples", C.c_uint32), ("depth", C.c_uint32),
                ("gamma", C.c_float), ("mint", C.c_float), ("maxt", C.c_float),
                ("integrator", C.c_uint32), ("sampler", C.c_uint32), ("accel", C.c_uint32), ("flags", C.c_uint32),
                ("seed", C.c_uint64),
                ("row_block", C.c_uint32), ("part_index", C.c_uint32), ("part_count", C.c_uint32), ("reserved", C.c_uint32)]


class RtwStats(C.Structure):
    _fields_ = [("camera_rays", C.c_uint64), ("segments", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("node_tests", C.c_uint64), ("nan_pixels", C.c_uint32), ("rows", C.c_uint32),
                ("kernel_ms", C.c_float), ("total_ms", C.c_float),
                ("phase_steps", C.c_uint64 * 6), ("phase_lanes", C.c_uint64 * 6), ("quad_tests", C.c_uint64),
                ("enqueue_ms", C.c_float), ("start_ms", C.c_float)]

    def as_dict(self):
        return {k: (list(getattr(self, k)) if k.startswith("phase_") else getattr(self, k)) for k, _ in self._fields_}


_lib = None


def lib() -> C.CDLL:
    """Load librtw_hip.so.  Fails loudly: there is no other implementation to fall back to."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          f"(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    fp = C.POINTER(C.c_float)
    L.rtw_abi_version.restype = C.c_int
    L.rtw_device_count.restype = C.c_int
    L.rtw_strerror.restype = C.c_char_p
    L.rtw_strerror.argtypes = [C.c_int]
    L.rtw_last_hip_error.restype = C.c_int
    L.rtw_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.rtw_ctx_destroy.argtypes = [C.c_void_p]
    L.rtw_ctx_destroy.restype = None
    L.rtw_ctx_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.rtw_ctx_set_scene.argtypes = [C.c_void_p, C.POINTER(RtwScene), C.c_float, C.c_float]
    L.rtw_ctx_render.argtypes = [C.c_void_p, C.POINTER(RtwCamera), C.POINTER(RtwParams), C.c_void_p, C.POINTER(RtwStats)]
    L.rtw_ctx_render_multi.argtypes = [C.c_void_p, C.POINTER(RtwCamera), C.POINTER(RtwParams), C.c_float, C.c_uint32, C.c_uint32,
                                       C.c_void_p, C.POINTER(RtwStats)]
    L.rtw_render.argtypes = [C.POINTER(RtwCamera), C.POINTER(RtwScene), C.POINTER(RtwParams), C.c_void_p, C.POINTER(RtwStats)]
    L.rtw_ctx_set_option.argtypes = [C.c_void_p, C.c_uint32, C.c_double]
    L.rtw_mgpu_create.argtypes = [C.POINTER(C.c_int), C.c_uint32, C.POINTER(C.c_void_p)]
    L.rtw_mgpu_destroy.argtypes = [C.c_void_p]
    L.rtw_mgpu_destroy.restype = None
    L.rtw_mgpu_set_scene.argtypes = [C.c_void_p, C.POINTER(RtwScene), C.c_float, C.c_float]
    L.rtw_mgpu_set_option.argtypes = [C.c_void_p, C.c_uint32, C.c_double]
    L.rtw_mgpu_render.argtypes = [C.c_void_p, C.POINTER(RtwCamera), C.POINTER(RtwParams), C.c_void_p, C.POINTER(RtwStats), C.POINTER(RtwStats)]
    L.rtw_render_multi_gpu.argtypes = [C.POINTER(C.c_int), C.c_uint32, C.POINTER(RtwCamera), C.POINTER(RtwScene), C.POINTER(RtwParams),
                                       C.c_void_p, C.POINTER(RtwStats)]
    L.rtw_viewport_new.argtypes = [C.c_uint32, C.c_float, fp, fp, fp, fp, fp, C.POINTER(RtwCamera), C.POINTER(C.c_uint32)]
    L.rtw_viewport_new_from_res.argtypes = [C.c_uint32, C.c_uint32, fp, fp, fp, fp, fp, C.POINTER(RtwCamera), C.POINTER(C.c_uint32)]
    L.rtw_sphere_new.argtypes = [fp, C.c_float, fp, fp, fp, C.POINTER(RtwSphere)]
    L.rtw_sphere_new_with_texture.argtypes = [fp, C.c_float, fp, fp, fp, C.c_int32, C.POINTER(RtwSphere)]
    L.rtw_vec3_rotated.restype = None
    L.rtw_vec3_rotated.argtypes = [fp, fp, fp]
    L.rtw_tile_order.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(RtwCamera), C.POINTER(RtwScene), C.POINTER(C.c_uint32), C.c_uint32]
    L.rtw_part_rows.restype = C.c_uint32
    L.rtw_part_rows.argtypes = [C.c_uint32] * 4
    L.rtw_quantize_u8.restype = None
    L.rtw_quantize_u8.argtypes = [fp, C.c_size_t, C.POINTER(C.c_uint8)]
    L.rtw_camera2_new.argtypes = [C.c_float, fp, fp, fp, C.c_float, C.c_float, C.POINTER(RtwCamera)]
    L.rtw_quantize_u8_rust2.restype = None
    L.rtw_quantize_u8_rust2.argtypes = [fp, C.c_size_t, C.POINTER(C.c_uint8)]
    L.rtw_scene_to_json.restype = C.c_size_t
    L.rtw_scene_to_json.argtypes = [C.POINTER(RtwScene), C.c_char_p, C.c_size_t]
    L.rtw_scene_from_json.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(RtwSphere), C.c_uint32, C.POINTER(C.c_uint32),
                                      C.POINTER(RtwTexture), C.c_uint32, C.POINTER(C.c_uint32), fp, C.c_uint32, C.POINTER(C.c_uint32)]
    L.rtw_write_png_f32.argtypes = [C.c_char_p, fp, C.c_uint32, C.c_uint32]
    L.rtw_write_ppm_f32.argtypes = [C.c_char_p, fp, C.c_uint32, C.c_uint32]
    L.rtw_bvh_validate.argtypes = [C.POINTER(RtwScene), C.c_float, C.c_float] + [C.POINTER(C.c_uint32)] * 4
    L.rtw_scene_generate.argtypes = [C.c_uint32, C.c_uint64, C.POINTER(RtwSphere), C.c_uint32, C.POINTER(C.c_uint32),
                                     C.POINTER(RtwTexture), C.c_uint32, C.POINTER(C.c_uint32),
                                     fp, C.c_uint32, C.POINTER(C.c_uint32)]
    L.rtw_scene_default_view.argtypes = [C.c_uint32, C.POINTER(RtwCamera), C.POINTER(RtwParams)]
    L.rtw_quad_new.argtypes = [fp, fp, fp, fp, fp, fp, C.POINTER(RtwQuad)]
    L.rtw_box_quads.argtypes = [fp, fp, fp, fp, C.POINTER(RtwQuad)]
    L.rtw_scene_generate_geom.argtypes = [C.c_uint32, C.c_uint64, C.POINTER(RtwSphere), C.POINTER(RtwQuad), C.POINTER(RtwInstance),
                                          C.POINTER(RtwSphere), C.POINTER(RtwQuad), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), fp]
    _lib = L
    return L


def _strerror(status: int) -> str:
    try:
        return lib().rtw_strerror(status).decode()
    except Exception:  # library itself missing
        return "?"


def _check(status: int, what: str):
    if status != RTW_OK:
        raise RtwError(status, what)


def _f3(v) -> Optional[C.Array]:
    if v is None:
        return None
    return (C.c_float * 3)(*[float(x) for x in v])


def _fptr(arr):
    return None if arr is None else C.cast(arr, C.POINTER(C.c_float))


def _f1(v):
    return None if v is None else C.pointer(C.c_float(float(v)))


# ---- reference-shaped host objects ---------------------------------------------------------------
class Sphere:
    """`Sphere` of Rust/src/objects/sphere.rs:13-20 (constructors :151-247)."""

    def __init__(self, pod: RtwSphere):
        self.pod = pod

    @staticmethod
    def new(origin, r, col_mod=None, mat=None) -> "Sphere":
        s = RtwSphere()
        _check(lib().rtw_sphere_new(_fptr(_f3(origin)), float(r), _fptr(_f3(col_mod)), _fptr(_f3(mat)), None, C.byref(s)), "rtw_sphere_new")
        return Sphere(s)

    @staticmethod
    def new_moving(origin, r, col_mod, mat, velocity) -> "Sphere":
        s = RtwSphere()
        _check(lib().rtw_sphere_new(_fptr(_f3(origin)), float(r), _fptr(_f3(col_mod)), _fptr(_f3(mat)), _fptr(_f3(velocity)), C.byref(s)), "rtw_sphere_new")
        return Sphere(s)

    @staticmethod
    def new_with_texture(origin, r, col_mod, mat, tex_index: int, velocity=None) -> "Sphere":
        s = RtwSphere()
        _check(lib().rtw_sphere_new_with_texture(_fptr(_f3(origin)), float(r), _fptr(_f3(col_mod)), _fptr(_f3(mat)),
                                                 _fptr(_f3(velocity)), int(tex_index), C.byref(s)), "rtw_sphere_new_with_texture")
        return Sphere(s)

    @staticmethod
    def with_albedo(origin, r, albedo, mat=None, velocity=None) -> "Sphere":
        """Albedo exactly `albedo` (texture = albedo, col_mod = 1): opts out of Sphere::new's c*c quirk."""
        sp = Sphere.new_moving(origin, r, (1.0, 1.0, 1.0), mat, velocity or (0.0, 0.0, 0.0))
        for k in range(3):
            sp.pod.tex_color[k] = float(albedo[k])
        return sp


class Quad:
    """`Quad` of Rust/src/objects/quad.rs:8-20; `new` is :84-110 with ImageTexture::from_color(color)."""

    def __init__(self, pod: RtwQuad):
        self.pod = pod

    @staticmethod
    def new(origin, u, v, mat=None, color=(1.0, 1.0, 1.0), emitted=None, tex_index: int = -1) -> "Quad":
        q = RtwQuad()
        _check(lib().rtw_quad_new(_fptr(_f3(origin)), _fptr(_f3(u)), _fptr(_f3(v)), _fptr(_f3(mat)), _fptr(_f3(emitted)),
                                  _fptr(_f3(color)), C.byref(q)), "rtw_quad_new")
        q.tex = int(tex_index)
        return Quad(q)


class Instance:
    """`Instance` of Rust/src/objects/instance.rs:27-38: member spheres and quads, translation, Euler rotation,
    optional constant-density medium (`dist_fn = &const_density; density = d`)."""

    def __init__(self, spheres: Sequence = (), quads: Sequence = ()):
        self.spheres = [s.pod if isinstance(s, Sphere) else s for s in spheres]
        self.quads = [q.pod if isinstance(q, Quad) else q for q in quads]
        self.translation = [0.0, 0.0, 0.0]
        self.rotation = [0.0, 0.0, 0.0]
        self.density, self.medium = 0.0, MEDIUM_SURFACE

    @staticmethod
    def new(spheres, quads) -> "Instance":
        return Instance(spheres, quads)

    @staticmethod
    def new_sphere(spheres) -> "Instance":
        return Instance(spheres, ())

    @staticmethod
    def new_quads(quads) -> "Instance":
        return Instance((), quads)

    @staticmethod
    def new_box(a, b, color, mat) -> "Instance":
        """Instance::new_box (instance.rs:83-176) with ImageTexture::from_color(color)."""
        q = (RtwQuad * 6)()
        _check(lib().rtw_box_quads(_fptr(_f3(a)), _fptr(_f3(b)), _fptr(_f3(mat)), _fptr(_f3(color)), q), "rtw_box_quads")
        return Instance((), [RtwQuad.from_buffer_copy(x) for x in q])

    def translate(self, vec):                      # instance.rs:241-243 (f32 `+=`)
        self.translation = [float(np.float32(a) + np.float32(b)) for a, b in zip(self.translation, vec)]

    def rotate(self, rot):                         # instance.rs:234-236
        self.rotation = [float(np.float32(a) + np.float32(b)) for a, b in zip(self.rotation, rot)]

    def const_density(self, density: float):       # `x.dist_fn = &const_density; x.density = d` (main.rs:355-356)
        self.density, self.medium = float(density), MEDIUM_CONST_DENSITY


class Scene:
    """`Scene` (Rust/src/viewport.rs:79-151): spheres (+ image textures), quads, instances, background colour."""

    def __init__(self, spheres: Sequence, textures: Sequence[np.ndarray] = (), background=(0.0, 0.0, 0.0),
                 quads: Sequence = (), instances: Sequence = (), emission_images=None):
        """`emission_images` {texture index: index of the texture that is its Rust2 `emmit_img`} (Rust2/src/objects/texture.rs:34-41;
        RTW_INTEGRATOR_RUST2 only)."""
        pods = [s.pod if isinstance(s, Sphere) else s for s in spheres]
        self._spheres = (RtwSphere * max(1, len(pods)))(*pods)
        self.n_spheres = len(pods)
        descs, flat, off = [], [], 0
        for img in textures:                      # img: [col(height)][row(width)][3] float32
            img = np.ascontiguousarray(img, dtype=np.float32)
            h, w = img.shape[0], img.shape[1]
            descs.append(RtwTexture(w, h, off, 0))
            flat.append(img.reshape(-1, 3))
            off += w * h
        for t, e in (emission_images or {}).items():
            descs[t].emit_tex = int(e) + 1
        self._textures = (RtwTexture * max(1, len(descs)))(*descs)
        self.n_textures = len(descs)
        self._texels = np.concatenate(flat, axis=0).astype(np.float32) if flat else np.zeros((1, 3), np.float32)
        self.n_texels = off
        self.pod = RtwScene()
        self.pod.spheres = C.cast(self._spheres, C.POINTER(RtwSphere))
        self.pod.textures = C.cast(self._textures, C.POINTER(RtwTexture))
        self.pod.texels = self._texels.ctypes.data_as(C.POINTER(C.c_float))
        self.pod.n_spheres, self.pod.n_textures, self.pod.n_texels = self.n_spheres, self.n_textures, self.n_texels
        for k in range(3):
            self.pod.background[k] = float(background[k])
        self._set_geom([q.pod if isinstance(q, Quad) else q for q in quads], list(instances))

    def _set_geom(self, quads, instances):
        """Flatten quads and instances into the ABI's pools (RtwScene.quads / instances / inst_spheres / inst_quads)."""
        isph, iquad, inst = [], [], []
        for it in instances:
            if isinstance(it, Instance):
                r = RtwInstance(len(isph), len(it.spheres), len(iquad), len(it.quads))
                for k in range(3):
                    r.translation[k], r.rotation[k] = it.translation[k], it.rotation[k]
                r.density, r.medium = it.density, it.medium
                isph += it.spheres
                iquad += it.quads
                inst.append(r)
            else:                                   # (RtwInstance, member spheres, member quads) with absolute ranges
                inst.append(it)
        self._install_geom(quads, inst, isph, iquad)

    def _install_geom(self, quads, inst, isph, iquad):
        self._quads = (RtwQuad * max(1, len(quads)))(*quads)
        self._instances = (RtwInstance * max(1, len(inst)))(*inst)
        self._inst_spheres = (RtwSphere * max(1, len(isph)))(*isph)
        self._inst_quads = (RtwQuad * max(1, len(iquad)))(*iquad)
        self.pod.quads = C.cast(self._quads, C.POINTER(RtwQuad))
        self.pod.instances = C.cast(self._instances, C.POINTER(RtwInstance))
        self.pod.inst_spheres = C.cast(self._inst_spheres, C.POINTER(RtwSphere))
        self.pod.inst_quads = C.cast(self._inst_quads, C.POINTER(RtwQuad))
        self.pod.n_quads, self.pod.n_instances = len(quads), len(inst)
        self.pod.n_inst_spheres, self.pod.n_inst_quads = len(isph), len(iquad)
        self.n_quads, self.n_instances = len(quads), len(inst)

    @staticmethod
    def new_sphere(spheres: Sequence) -> "Scene":
        return Scene(spheres)

    @staticmethod
    def new_quad(quads: Sequence) -> "Scene":
        """Scene::new_quad (viewport.rs:106-121)."""
        return Scene((), quads=quads)

    @staticmethod
    def new(spheres: Sequence, quads: Sequence, instances: Sequence) -> "Scene":
        """Scene::new(spheres, quads, instances) (viewport.rs:122-135); background_color is black."""
        return Scene(spheres, quads=quads, instances=instances)

    @staticmethod
    def generate_geom(which: int, scene_seed: int = 42) -> "Scene":
        """A scene with quads / instances laid out by the host library (SCENE_QUAD_TEST, SCENE_PRESENTATION)."""
        L = lib()
        counts = (C.c_uint32 * 5)()
        bg = (C.c_float * 3)()
        _check(L.rtw_scene_generate_geom(which, scene_seed, None, None, None, None, None, None, counts, bg), "rtw_scene_generate_geom")
        n = [int(x) for x in counts]
        sp, qd, ins = (RtwSphere * max(1, n[0]))(), (RtwQuad * max(1, n[1]))(), (RtwInstance * max(1, n[2]))()
        isp, iqd = (RtwSphere * max(1, n[3]))(), (RtwQuad * max(1, n[4]))()
        caps = (C.c_uint32 * 5)(*n)
        _check(L.rtw_scene_generate_geom(which, scene_seed, sp, qd, ins, isp, iqd, caps, counts, bg), "rtw_scene_generate_geom")
        sc = Scene(list(sp)[:n[0]], background=tuple(bg))
        sc._install_geom(list(qd)[:n[1]], list(ins)[:n[2]], list(isp)[:n[3]], list(iqd)[:n[4]])
        return sc

    def to_json(self) -> str:
        """`Into<JsonValue> for Scene` (Rust/src/viewport.rs:174-180)."""
        n = lib().rtw_scene_to_json(C.byref(self.pod), None, 0)
        buf = C.create_string_buffer(n + 1)
        lib().rtw_scene_to_json(C.byref(self.pod), buf, n + 1)
        return buf.value.decode()

    @staticmethod
    def from_json(text: str) -> "Scene":
        """`TryFrom<JsonValue> for Scene` (Rust/src/viewport.rs:181-205); raises RtwError like ParseError."""
        L = lib()
        raw = text.encode()
        ns, nt, nx = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(L.rtw_scene_from_json(raw, len(raw), None, 0, C.byref(ns), None, 0, C.byref(nt), None, 0, C.byref(nx)), "rtw_scene_from_json")
        sp = (RtwSphere * max(1, ns.value))()
        tx = (RtwTexture * max(1, nt.value))()
        tl = np.zeros((max(1, nx.value), 3), np.float32)
        _check(L.rtw_scene_from_json(raw, len(raw), sp, ns.value, C.byref(ns), tx, nt.value, C.byref(nt),
                                     tl.ctypes.data_as(C.POINTER(C.c_float)), nx.value, C.byref(nx)), "rtw_scene_from_json")
        return Scene._from_arrays(sp, ns.value, tx, nt.value, tl, nx.value)

    @staticmethod
    def _from_arrays(sp, ns, tx, nt, tl, nx) -> "Scene":
        sc = Scene(list(sp)[:ns])
        sc._textures, sc.n_textures, sc._texels, sc.n_texels = tx, nt, tl, nx
        sc.pod.textures = C.cast(tx, C.POINTER(RtwTexture))
        sc.pod.texels = tl.ctypes.data_as(C.POINTER(C.c_float))
        sc.pod.n_textures, sc.pod.n_texels = nt, nx
        return sc

    @staticmethod
    def generate(which: int, scene_seed: int = 42) -> "Scene":
        """One of the BASELINE config scenes, laid out by the C++ host library."""
        L = lib()
        ns, nt, nx = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(L.rtw_scene_generate(which, scene_seed, None, 0, C.byref(ns), None, 0, C.byref(nt), None, 0, C.byref(nx)), "rtw_scene_generate")
        sp = (RtwSphere * max(1, ns.value))()
        tx = (RtwTexture * max(1, nt.value))()
        tl = np.zeros((max(1, nx.value), 3), np.float32)
        _check(L.rtw_scene_generate(which, scene_seed, sp, ns.value, C.byref(ns), tx, nt.value, C.byref(nt),
                                    tl.ctypes.data_as(C.POINTER(C.c_float)), nx.value, C.byref(nx)), "rtw_scene_generate")
        return Scene._from_arrays(sp, ns.value, tx, nt.value, tl, nx.value)


class Viewport:
    """`Viewport` of Rust/src/viewport.rs:49-77; `new`/`new_from_res` are :308-428.

    `render(ray_color, scene)` takes the integrator as an enum (a host closure cannot run on the GPU):
    INTEGRATOR_GRADIENT == ray_color_gradient, INTEGRATOR_BG_COLOR == ray_color_bg_color.
    """

    def __init__(self, cam: RtwCamera, width: int, height: int, samples: int, depth: int, gamma: float):
        self.cam, self.width, self.height = cam, width, height
        self.samples, self.depth, self.gamma = samples, depth, gamma
        self.shutter_speed, self.fps, self.frame = 0.0, 30.0, 0
        self.seed = 1
        self.mint, self.maxt = 0.001, 100000.0

    @staticmethod
    def new(width, aspect_ratio, samples, depth, gamma, vfov=None, origin=None, direction=None, vup=None, msg=None, lens_radius=None):
        cam, h = RtwCamera(), C.c_uint32()
        _check(lib().rtw_viewport_new(int(width), float(aspect_ratio), _f1(vfov), _fptr(_f3(origin)), _fptr(_f3(direction)),
                                      _fptr(_f3(vup)), _f1(lens_radius), C.byref(cam), C.byref(h)), "rtw_viewport_new")
        return Viewport(cam, int(width), h.value, samples, depth, gamma)

    @staticmethod
    def new_from_res(width, height, samples, depth, gamma, vfov=None, origin=None, direction=None, vup=None, msg=None, lens_radius=None):
        cam, h = RtwCamera(), C.c_uint32()
        _check(lib().rtw_viewport_new_from_res(int(width), int(height), _f1(vfov), _fptr(_f3(origin)), _fptr(_f3(direction)),
                                               _fptr(_f3(vup)), _f1(lens_radius), C.byref(cam), C.byref(h)), "rtw_viewport_new_from_res")
        return Viewport(cam, int(width), h.value, samples, depth, gamma)

    def params(self, integrator=INTEGRATOR_GRADIENT, sampler=SAMPLER_STRATIFIED, accel=ACCEL_BVH) -> RtwParams:
        p = RtwParams()
        p.width, p.height, p.samples, p.depth = self.width, self.height, self.samples, self.depth
        p.gamma, p.mint, p.maxt = self.gamma, self.mint, self.maxt
        p.integrator, p.sampler, p.accel, p.flags, p.seed = integrator, sampler, accel, 0, self.seed
        p.row_block, p.part_index, p.part_count = 8, 0, 1
        return p

    def camera(self) -> RtwCamera:
        cam = RtwCamera.from_buffer_copy(self.cam)
        cam.time0 = float(np.float32(self.frame) / np.float32(self.fps))    # viewport.rs:279
        cam.shutter = float(self.shutter_speed)
        return cam

    def render(self, ray_color: int, scene: Scene, device: int = 0, accel: int = ACCEL_BVH) -> np.ndarray:
        """Viewport::render (stratified, serial in the reference; viewport.rs:430-478) -> [H][W][3] f32."""
        return self._render(ray_color, SAMPLER_STRATIFIED, scene, device, accel)

    def async_render(self, ray_color: int, scene: Scene, device: int = 0, accel: int = ACCEL_BVH) -> np.ndarray:
        """async_render / render_row (viewport.rs:215-305): exactly `samples` rays, shutter time."""
        return self._render(ray_color, SAMPLER_ROW, scene, device, accel)

    def render_multi(self, ray_color: int, scene: Scene, device: int = 0, accel: int = ACCEL_BVH):
        """render_multi (viewport.rs:249-269): frames start_frame .. start_frame + number_of_frames, each rendered
        with time = frame / fps (the scene and its time-expanded BVH are uploaded once for the whole clip)."""
        start = getattr(self, "start_frame", 0)
        count = getattr(self, "number_of_frames", 1)
        with Renderer(device) as r:
            t0 = float(np.float32(start) / np.float32(self.fps))
            t1 = float(np.float32(start + max(count, 1) - 1) / np.float32(self.fps)) + float(self.shutter_speed)
            r.set_scene(scene, t0, t1)
            p = self.params(ray_color, SAMPLER_ROW, accel)
            video = np.empty((max(count, 0), self.height, self.width, 3), np.float32)
            st = (RtwStats * max(count, 1))()
            cam = self.camera()
            _check(lib().rtw_ctx_render_multi(r._h, C.byref(cam), C.byref(p), float(self.fps), int(start), int(count),
                                              C.c_void_p(video.ctypes.data), st), "rtw_ctx_render_multi")
        self.frame = start + max(count, 1) - 1
        return [video[i] for i in range(count)]

    def render_no_rand(self, ray_color: int, scene: Scene, device: int = 0, accel: int = ACCEL_BVH) -> np.ndarray:
        return self._render(ray_color, SAMPLER_NO_RAND, scene, device, accel)

    def _render(self, ray_color, sampler, scene, device, accel):
        with Renderer(device) as r:
            cam = self.camera()
            r.set_scene(scene, cam.time0, cam.time0 + cam.shutter)
            img, _ = r.render(cam, self.params(ray_color, sampler, accel))
        return img


class Renderer:
    """One `rtw_ctx`: one GPU, one stream, device-resident scene + BVH."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        _check(lib().rtw_ctx_create(int(device), C.byref(self._h)), "rtw_ctx_create")
        self._scene = None

    def close(self):
        if self._h:
            lib().rtw_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream: int):
        _check(lib().rtw_ctx_set_stream(self._h, C.c_void_p(hip_stream)), "rtw_ctx_set_stream")

    def set_scene(self, scene: Scene, t_begin: float = 0.0, t_end: float = 0.0):
        self._scene = scene     # keep host arrays alive
        _check(lib().rtw_ctx_set_scene(self._h, C.byref(scene.pod), float(t_begin), float(t_end)), "rtw_ctx_set_scene")

    def set_option(self, key: int, value: float):
        """Tuning knobs (OPT_*); none of them changes the image."""
        _check(lib().rtw_ctx_set_option(self._h, int(key), float(value)), "rtw_ctx_set_option")

    def render(self, cam: RtwCamera, params: RtwParams, out=None):
        """Render into `out`: None -> new numpy array; numpy array -> host buffer; int -> raw device pointer
        (e.g. torch_tensor.data_ptr()) of [rows][width][3] f32.  Returns (out, RtwStats)."""
        rows = lib().rtw_part_rows(params.height, params.row_block, params.part_index, params.part_count)
        st = RtwStats()
        if out is None:
            out = np.empty((rows, params.width, 3), np.float32)
        if isinstance(out, np.ndarray):
            assert out.dtype == np.float32 and out.flags["C_CONTIGUOUS"] and out.size == rows * params.width * 3
            ptr = C.c_void_p(out.ctypes.data)
        else:
            ptr = C.c_void_p(int(out))
        _check(lib().rtw_ctx_render(self._h, C.byref(cam), C.byref(params), ptr, C.byref(st)), "rtw_ctx_render")
        return out, st


class MultiRenderer:
    """`rtw_mgpu`: one frame over several GPUs of a node from ONE process -- the fork / ordered join of the reference's row
    tasks (Rust/src/viewport.rs:236-244).  `devices` are HIP ordinals and may repeat."""

    def __init__(self, devices: Sequence[int]):
        self._h = C.c_void_p()
        self.n = len(devices)
        arr = (C.c_int * max(1, self.n))(*[int(d) for d in devices])
        _check(lib().rtw_mgpu_create(arr, self.n, C.byref(self._h)), "rtw_mgpu_create")
        self._scene = None

    def close(self):
        if self._h:
            lib().rtw_mgpu_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_scene(self, scene: Scene, t_begin: float = 0.0, t_end: float = 0.0):
        self._scene = scene
        _check(lib().rtw_mgpu_set_scene(self._h, C.byref(scene.pod), float(t_begin), float(t_end)), "rtw_mgpu_set_scene")

    def set_option(self, key: int, value: float):
        _check(lib().rtw_mgpu_set_option(self._h, int(key), float(value)), "rtw_mgpu_set_option")

    def render(self, cam: RtwCamera, params: RtwParams, out=None):
        """Full frame into `out` (None -> new numpy array; numpy array; int -> raw device pointer).
        Returns (out, total RtwStats, [per-device RtwStats])."""
        if out is None:
            out = np.empty((params.height, params.width, 3), np.float32)
        if isinstance(out, np.ndarray):
            assert out.dtype == np.float32 and out.flags["C_CONTIGUOUS"] and out.size == params.height * params.width * 3
            ptr = C.c_void_p(out.ctypes.data)
        else:
            ptr = C.c_void_p(int(out))
        per = (RtwStats * max(1, self.n))()
        tot = RtwStats()
        _check(lib().rtw_mgpu_render(self._h, C.byref(cam), C.byref(params), ptr, per, C.byref(tot)), "rtw_mgpu_render")
        return out, tot, list(per)[: self.n]


def default_view(which: int):
    cam, p = RtwCamera(), RtwParams()
    _check(lib().rtw_scene_default_view(which, C.byref(cam), C.byref(p)), "rtw_scene_default_view")
    return cam, p


def quantize_u8(img: np.ndarray) -> np.ndarray:
    """write_img_f32's 8-bit rule (Rust/src/write_img.rs:11-15)."""
    a = np.ascontiguousarray(img, np.float32)
    out = np.empty(a.shape, np.uint8)
    lib().rtw_quantize_u8(a.ctypes.data_as(C.POINTER(C.c_float)), a.size, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def camera2_new(aspect, origin, vup, direction, vfov, lens_radius) -> RtwCamera:
    """Rust2 `Camera::new` (Rust2/src/viewport/camera.rs:19-53), for SAMPLER_CENTRES."""
    cam = RtwCamera()
    _check(lib().rtw_camera2_new(float(aspect), _fptr(_f3(origin)), _fptr(_f3(vup)), _fptr(_f3(direction)), float(vfov),
                                 float(lens_radius), C.byref(cam)), "rtw_camera2_new")
    return cam


def quantize_u8_rust2(img: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(img, np.float32)
    out = np.empty(a.shape, np.uint8)
    lib().rtw_quantize_u8_rust2(a.ctypes.data_as(C.POINTER(C.c_float)), a.size, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def write_img_f32(img: np.ndarray, filename: str):
    """write_img_f32 (Rust/src/write_img.rs:6-19): 8-bit RGB PNG."""
    a = np.ascontiguousarray(img, np.float32)
    _check(lib().rtw_write_png_f32(filename.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[1], a.shape[0]), "rtw_write_png_f32")


def write_ppm(filename: str, img: np.ndarray):
    """write_ppm (C++/src/ppm_writer.cpp:3-27): P3 text."""
    a = np.ascontiguousarray(img, np.float32)
    _check(lib().rtw_write_ppm_f32(filename.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[1], a.shape[0]), "rtw_write_ppm_f32")


def vec3_rotated(v, rot) -> np.ndarray:
    """Vec3::rotated (Rust/src/vec3.rs:161-181)."""
    out = (C.c_float * 3)()
    lib().rtw_vec3_rotated(_fptr(_f3(v)), _fptr(_f3(rot)), out)
    return np.array(list(out), np.float32)


def device_count() -> int:
    return int(lib().rtw_device_count())
