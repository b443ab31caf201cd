"""ctypes binding of the CPU oracle (oracle/librtw_oracle.so) and of the reference-object checker
(oracle/_ref/librtw_ref.so).  TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

import rtw_amd as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.environ.get("RTW_ORACLE_LIB", os.path.join(ORACLE_DIR, "librtw_oracle.so"))   # override: the sanitizer build
REF_SO = os.path.join(ORACLE_DIR, "_ref", "librtw_ref.so")


class Bounce(C.Structure):
    _fields_ = [("hit", C.c_int32), ("sphere", C.c_int32), ("front_face", C.c_int32), ("cannot_refract", C.c_int32),
                ("t", C.c_float), ("ratio", C.c_float), ("normal", C.c_float * 3), ("point", C.c_float * 3),
                ("unit_dir", C.c_float * 3), ("next_dir", C.c_float * 3)]


_lib = None
_ref = None


def build():
    """gcc the oracle (and, only where /root/reference exists, the reference objects)."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True, capture_output=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build()
        L = C.CDLL(ORACLE_SO)
        fp = C.POINTER(C.c_float)
        L.rtw_oracle_render.argtypes = [C.POINTER(R.RtwCamera), C.POINTER(R.RtwScene), C.POINTER(R.RtwParams), fp,
                                        C.POINTER(R.RtwStats), C.c_int]
        L.rtw_oracle_viewport_new.argtypes = [C.c_uint32, C.c_float, fp, fp, fp, fp, fp, C.POINTER(R.RtwCamera), C.POINTER(C.c_uint32)]
        L.rtw_oracle_trace_ray.argtypes = [fp, fp, C.c_float, C.POINTER(R.RtwScene), C.POINTER(R.RtwParams), C.c_uint32,
                                           C.c_uint32, C.POINTER(Bounce), C.c_int, fp]
        L.rtw_oracle_rng_seed.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        L.rtw_oracle_rng_seed.restype = None
        L.rtw_oracle_rng_next.argtypes = [C.POINTER(C.c_uint32)]
        L.rtw_oracle_rng_next.restype = C.c_float
        L.rtw_oracle_rust2_texel_index.argtypes = [C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_int]
        L.rtw_oracle_rust2_texel_index.restype = C.c_uint32
        L.rtw_oracle_sphere_uv.argtypes = [fp, C.c_size_t, C.c_int, fp]
        L.rtw_oracle_sphere_uv.restype = None
        L.rtw_oracle_rotated.argtypes = [fp, fp, fp]
        L.rtw_oracle_rotated.restype = None
        _lib = L
    return _lib


def rotated(v, rot):
    out = (C.c_float * 3)()
    lib().rtw_oracle_rotated((C.c_float * 3)(*v), (C.c_float * 3)(*rot), out)
    return np.array(list(out), np.float32)


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        fp = C.POINTER(C.c_float)
        L.rtw_ref_sphere_hit.argtypes = [fp, C.c_float, fp, fp, fp, C.c_float, C.c_float, C.c_uint, C.POINTER(C.c_double)]
        L.rtw_ref_camera.argtypes = [C.c_uint32, C.c_float, C.c_float, fp, fp, fp, C.c_float, C.POINTER(R.RtwCamera), C.POINTER(C.c_uint32)]
        L.rtw_ref_s_test.argtypes = [C.POINTER(C.POINTER(C.c_char)), C.POINTER(C.c_size_t), C.POINTER(C.POINTER(C.c_char)), C.POINTER(C.c_size_t)]
        L.rtw_ref_free.argtypes = [C.c_void_p]
        L.rtw_ref_free.restype = None
        L.rtw_ref_render.argtypes = [C.POINTER(R.RtwCamera), C.POINTER(R.RtwScene), C.POINTER(R.RtwParams), C.c_uint,
                                     C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        _ref = L
    return _ref


def render(cam, scene, params, threads=8):
    """rtw_oracle_render -> ([rows][W][3] f32, RtwStats)."""
    rows = R.lib().rtw_part_rows(params.height, params.row_block, params.part_index, params.part_count)
    out = np.empty((rows, params.width, 3), np.float32)
    st = R.RtwStats()
    rc = lib().rtw_oracle_render(C.byref(cam), C.byref(scene.pod), C.byref(params), out.ctypes.data_as(C.POINTER(C.c_float)),
                                 C.byref(st), threads)
    assert rc == 0, rc
    return out, st


def trace_ray(origin, direction, time, scene, params, pixel=0, sample=0, cap=64):
    buf = (Bounce * cap)()
    rgb = (C.c_float * 3)()
    o = (C.c_float * 3)(*origin)
    d = (C.c_float * 3)(*direction)
    n = lib().rtw_oracle_trace_ray(o, d, float(time), C.byref(scene.pod), C.byref(params), pixel, sample, buf, cap, rgb)
    assert n >= 0, n
    return list(buf)[:n], np.array(list(rgb), np.float32)


def viewport_new(width, aspect, vfov=None, origin=None, direction=None, vup=None, lens_radius=None):
    cam, h = R.RtwCamera(), C.c_uint32()
    rc = lib().rtw_oracle_viewport_new(width, aspect, R._f1(vfov), R._fptr(R._f3(origin)), R._fptr(R._f3(direction)),
                                       R._fptr(R._f3(vup)), R._f1(lens_radius), C.byref(cam), C.byref(h))
    assert rc == 0
    return cam, h.value


def ref_render(cam, scene, params, rand_seed=1):
    out = np.empty((params.height, params.width, 3), np.float64)
    seg, sec = C.c_uint64(), C.c_double()
    rc = ref().rtw_ref_render(C.byref(cam), C.byref(scene.pod), C.byref(params), rand_seed,
                              out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(seg), C.byref(sec))
    assert rc == 0
    return out, seg.value, sec.value
