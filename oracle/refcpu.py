"""ctypes binding of oracle/librefcpu.so -- the CPU checker (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module; the product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librefcpu.so")

RNG_LCG_TILES, RNG_LCG_SERIAL, RNG_COUNTER, RNG_CONST_U, RNG_COUNTER_DIRECT = 0, 1, 2, 3, 4


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("pos", Vec3), ("pixel00_loc", Vec3), ("pixel_delta_u", Vec3),
                ("pixel_delta_v", Vec3), ("defocus_disk_u", Vec3), ("defocus_disk_v", Vec3),
                ("defocus_angle", C.c_float)]


class CameraParams(C.Structure):
    _fields_ = [("pos", Vec3), ("target", Vec3), ("up", Vec3), ("vfov", C.c_float),
                ("defocus_angle", C.c_float), ("focus_dist", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("kernel_ms", C.c_double), ("total_ms", C.c_double), ("resolve_ms", C.c_double),
                ("chunk_spp", C.c_uint32), ("scan_tiles", C.c_uint32), ("wave_scans", C.c_uint64)]


class Opts(C.Structure):
    _fields_ = [("rng", C.c_int32), ("seed", C.c_uint32), ("sample_base", C.c_uint32),
                ("chunk_spp", C.c_int32), ("threads", C.c_int32), ("tile_rows", C.c_int32),
                ("part_index", C.c_int32), ("part_count", C.c_int32),
                ("hit_index_out", C.c_void_p), ("accum_out", C.c_void_p), ("stats", C.c_void_p)]


# numpy view of rt_sphere (40 bytes)
SPHERE_DTYPE = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("mat_type", "<i4"),
                         ("albedo", "<f4", 3), ("fuzz", "<f4"), ("ior", "<f4")])
assert SPHERE_DTYPE.itemsize == 40

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "librefcpu.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.refcpu_render.restype = C.c_int
        _lib.refcpu_render.argtypes = [C.POINTER(Camera), C.c_void_p, C.c_size_t, C.c_void_p,
                                       C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(Opts)]
        _lib.refcpu_camera_setup.restype = C.c_int
        _lib.refcpu_camera_setup.argtypes = [C.POINTER(CameraParams), C.c_uint32, C.c_uint32,
                                             C.POINTER(Camera)]
        _lib.refcpu_counter_key.restype = C.c_uint32
        _lib.refcpu_counter_key.argtypes = [C.c_uint32] * 3
        _lib.refcpu_counter_draw.restype = C.c_uint32
        _lib.refcpu_counter_draw.argtypes = [C.c_uint32] * 2
        _lib.refcpu_core_count.restype = C.c_int
    return _lib


def default_params():
    """init_camera's defaults, Main.c:915-922."""
    p = CameraParams()
    p.pos = Vec3(13.0, 2.0, 3.0)
    p.target = Vec3(0.0, 0.0, 0.0)
    p.up = Vec3(0.0, 1.0, 0.0)
    p.vfov = 60.0
    p.defocus_angle = 0.6
    p.focus_dist = 10.0
    return p


def height_for(width, aspect=16.0 / 9.0):
    """init_camera, Main.c:908: (u32)(window_width / aspect_ratio) in f32."""
    return int(np.float32(width) / np.float32(aspect))


def camera(width, height=None, params=None):
    if height is None:
        height = height_for(width)
    cam = Camera()
    p = params if params is not None else default_params()
    rc = lib().refcpu_camera_setup(C.byref(p), width, height, C.byref(cam))
    if rc != 0:
        raise RuntimeError("refcpu_camera_setup failed: %d" % rc)
    return cam


def camera_as_array(cam):
    return np.frombuffer(bytes(cam), dtype="<f4").copy()


def render(cam, spheres, width, height, spp, depth, rng=RNG_COUNTER, seed=0, sample_base=0,
           chunk_spp=0, threads=0, tile_rows=0, part_index=0, part_count=0,
           want_hit=False, want_accum=False):
    """Returns dict(rgba=HxWx4 u8, hit=HxW i32|None, accum=HxWx3 f32|None, stats=Stats)."""
    spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
    rgba = np.zeros((height, width, 4), dtype=np.uint8)
    hit = np.full((height, width), -2, dtype=np.int32) if want_hit else None
    acc = np.zeros((height, width, 3), dtype=np.float32) if want_accum else None
    st = Stats()
    o = Opts()
    o.rng, o.seed, o.sample_base, o.chunk_spp, o.threads = rng, seed, sample_base, chunk_spp, threads
    o.tile_rows, o.part_index, o.part_count = tile_rows, part_index, part_count
    o.hit_index_out = hit.ctypes.data if want_hit else None
    o.accum_out = acc.ctypes.data if want_accum else None
    o.stats = C.addressof(st)
    rc = lib().refcpu_render(C.byref(cam), spheres.ctypes.data, len(spheres), rgba.ctypes.data,
                             width, height, spp, depth, C.byref(o))
    if rc != 0:
        raise RuntimeError("refcpu_render failed: %d" % rc)
    return dict(rgba=rgba, hit=hit, accum=acc, stats=st)
