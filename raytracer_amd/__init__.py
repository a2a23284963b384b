"""raytracer_amd -- host-side Python mirror of the C ABI in include/rt.h.

The compute path is librtmi.so (hand-written HIP for gfx950).  This module only loads it
with ctypes and marshals numpy arrays / raw device pointers; there is no CPU fallback:
if the library is missing or no gfx950 device is present, calls raise.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTMI_LIB", os.path.join(HERE, "librtmi.so"))   # RTMI_LIB: diagnostic builds
CSRC = os.path.join(HERE, "csrc")

MODE_MC, MODE_CONST_U = 0, 1
BUILD_EXACT, BUILD_STRICT, BUILD_FAST = 0, 1, 2          # RT_BUILD_*
BUILDS = {"exact": BUILD_EXACT, "strict": BUILD_STRICT, "fast": BUILD_FAST}
CHUNK_AUTO = -1
MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC = 0, 1, 2


class RtError(RuntimeError):
    pass


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Camera(C.Structure):
    """rt_camera: what get_ray reads (Main.c:776-803)."""
    _fields_ = [("pos", Vec3), ("pixel00_loc", Vec3), ("pixel_delta_u", Vec3),
                ("pixel_delta_v", Vec3), ("defocus_disk_u", Vec3), ("defocus_disk_v", Vec3),
                ("defocus_angle", C.c_float)]


class CameraParams(C.Structure):
    """rt_camera_params: inputs of update_camera_view (Main.c:839-904)."""
    _fields_ = [("pos", Vec3), ("target", Vec3), ("up", Vec3), ("vfov", C.c_float),
                ("defocus_angle", C.c_float), ("focus_dist", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("kernel_ms", C.c_double), ("total_ms", C.c_double), ("resolve_ms", C.c_double),
                ("chunk_spp", C.c_uint32), ("scan_tiles", C.c_uint32), ("wave_scans", C.c_uint64),
                ("tile_products", C.c_uint64)]


class Opts(C.Structure):
    _fields_ = [("mode", C.c_int32), ("build", C.c_int32), ("seed", C.c_uint32),
                ("sample_base", C.c_uint32), ("chunk_spp", C.c_int32), ("tile_rows", C.c_int32),
                ("part_index", C.c_int32), ("part_count", C.c_int32),
                ("hit_index_out", C.c_void_p), ("accum_out", C.c_void_p), ("stats", C.c_void_p),
                ("rows", C.c_void_p), ("row_count", C.c_uint32)]


# numpy view of rt_sphere: sphere_t + material_t of the reference (Main.c:53-59, 74-79), 40 bytes
SPHERE_DTYPE = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("mat_type", "<i4"),
                         ("albedo", "<f4", 3), ("fuzz", "<f4"), ("ior", "<f4")])
assert SPHERE_DTYPE.itemsize == 40

EXPORTS = ["rt_init", "rt_init_devices", "rt_set_world", "rt_render", "rt_render_device", "rt_render_device_noalloc",
           "rt_render_progressive", "rt_part_rows", "rt_part_row_at", "rt_chunk_for", "rt_camera_setup", "rt_camera_defaults", "rt_height_for",
           "rt_scene_generate", "rt_world_layout", "rt_host_alloc", "rt_host_free", "rt_shutdown", "rt_last_error"]

_lib = None


def build(verbose=False):
    """Compile librtmi.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)


def lib():
    """Load librtmi.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RtError("%s not built; run raytracer_amd.build() or make -C %s" % (LIB_PATH, CSRC))
        L = C.CDLL(LIB_PATH)
        L.rt_init.restype = C.c_int
        L.rt_init.argtypes = [C.c_int]
        L.rt_init_devices.restype = C.c_int
        L.rt_init_devices.argtypes = [C.POINTER(C.c_int), C.c_int]
        L.rt_set_world.restype = C.c_int
        L.rt_set_world.argtypes = [C.c_void_p, C.c_size_t]
        for f in (L.rt_render, L.rt_render_device, L.rt_render_device_noalloc):
            f.restype = C.c_int
            f.argtypes = [C.POINTER(Camera), C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                          C.POINTER(Opts)]
        L.rt_render_progressive.restype = C.c_int
        L.rt_render_progressive.argtypes = [C.POINTER(Camera), C.c_void_p, C.c_uint32, C.c_uint32, C.c_int,
                                            C.c_int, C.POINTER(Opts), C.c_uint32]
        L.rt_part_rows.restype = C.c_uint32
        L.rt_part_rows.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int]
        L.rt_part_row_at.restype = C.c_uint32
        L.rt_part_row_at.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int]
        L.rt_chunk_for.restype = C.c_int
        L.rt_chunk_for.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int]
        L.rt_camera_setup.restype = C.c_int
        L.rt_camera_setup.argtypes = [C.POINTER(CameraParams), C.c_uint32, C.c_uint32, C.POINTER(Camera)]
        L.rt_camera_defaults.restype = None
        L.rt_camera_defaults.argtypes = [C.POINTER(CameraParams)]
        L.rt_height_for.restype = C.c_uint32
        L.rt_height_for.argtypes = [C.c_uint32, C.c_float]
        L.rt_world_layout.restype = C.c_int
        L.rt_world_layout.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.rt_scene_generate.restype = C.c_int
        L.rt_scene_generate.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.rt_host_alloc.restype = C.c_void_p
        L.rt_host_alloc.argtypes = [C.c_size_t]
        L.rt_host_free.restype = None
        L.rt_host_free.argtypes = [C.c_void_p]
        L.rt_shutdown.restype = None
        L.rt_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RtError("%s failed (%d): %s" % (what, rc, lib().rt_last_error().decode()))


def init(device=0):
    _check(lib().rt_init(device), "rt_init")


def init_devices(devices):
    """rt_init_devices: several GPUs in this one process (rt_render splits the rows among them)."""
    arr = (C.c_int * len(devices))(*devices)
    _check(lib().rt_init_devices(arr, len(devices)), "rt_init_devices")


def host_framebuffer(height, width):
    """rt_host_alloc: a page-locked RGBA8 framebuffer as a numpy array (DMA read-back).  Release it
    with host_free(array) when no longer used."""
    nbytes = height * width * 4
    p = lib().rt_host_alloc(nbytes)
    if not p:
        raise RtError("rt_host_alloc failed: %s" % lib().rt_last_error().decode())
    buf = (C.c_uint8 * nbytes).from_address(p)
    arr = np.frombuffer(buf, dtype=np.uint8).reshape(height, width, 4)
    arr[...] = 0
    return arr


def host_free(array):
    lib().rt_host_free(array.ctypes.data)


def shutdown():
    lib().rt_shutdown()


def set_world(spheres):
    spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
    _check(lib().rt_set_world(spheres.ctypes.data, len(spheres)), "rt_set_world")


def scene_generate(half_extent=11, lcg_seed=0):
    """rt_scene_generate: init_scene (Main.c:1212-1318) with a fixed evaluation order."""
    cap = 4 * half_extent * half_extent + 4
    out = np.zeros(cap, dtype=SPHERE_DTYPE)
    n = C.c_size_t(0)
    _check(lib().rt_scene_generate(half_extent, lcg_seed, out.ctypes.data, cap, C.byref(n)), "rt_scene_generate")
    return out[:n.value].copy()


def world_layout(spheres, split_tiles=True):
    """rt_world_layout: the device order of the spheres (order[k] = caller's index of the k-th sphere on the device)."""
    spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
    order = np.zeros(len(spheres), dtype=np.int32)
    _check(lib().rt_world_layout(spheres.ctypes.data, len(spheres), 1 if split_tiles else 0, order.ctypes.data), "rt_world_layout")
    return order


def default_params():
    p = CameraParams()
    lib().rt_camera_defaults(C.byref(p))
    return p


def height_for(width, aspect=16.0 / 9.0):
    return int(lib().rt_height_for(width, aspect))


def camera(width, height=None, params=None):
    """rt_camera for the reference's default view (or `params`) at width x height."""
    if height is None:
        height = height_for(width)
    p = params if params is not None else default_params()
    cam = Camera()
    _check(lib().rt_camera_setup(C.byref(p), width, height, C.byref(cam)), "rt_camera_setup")
    return cam


def camera_as_array(cam):
    return np.frombuffer(bytes(cam), dtype="<f4").copy()


def part_rows(height, tile_rows, part_index, part_count):
    return int(lib().rt_part_rows(height, tile_rows, part_index, part_count))


def part_row_map(height, tile_rows, part_index, part_count):
    """Image rows owned by a part, in the order of its compact buffer."""
    n = part_rows(height, tile_rows, part_index, part_count)
    return np.array([lib().rt_part_row_at(k, tile_rows, part_index, part_count) for k in range(n)],
                    dtype=np.int64)


def build_id(build=None, strict_fp=False):
    """RT_BUILD_* from a name ("exact", "strict", "fast"), a number, or -- when `build` is None -- from
    strict_fp: True = the strict build (reference samplers), False = the exact build (the default)."""
    if build is None:
        return BUILD_STRICT if strict_fp else BUILD_EXACT
    return BUILDS[build] if isinstance(build, str) else int(build)


def _opts(mode, strict_fp, seed, sample_base, chunk_spp, tile_rows, part_index, part_count, rows=None, build=None):
    o = Opts()
    if rows is not None:
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        o._rows_keepalive = rows
        o.rows, o.row_count = rows.ctypes.data, len(rows)
    o.mode, o.build, o.seed, o.sample_base = mode, build_id(build, strict_fp), seed, sample_base
    o.chunk_spp, o.tile_rows, o.part_index, o.part_count = chunk_spp, tile_rows, part_index, part_count
    return o


def render(cam, width, height, spp, max_depth, mode=MODE_MC, strict_fp=False, seed=0, sample_base=0,
           chunk_spp=0, tile_rows=0, part_index=0, part_count=0, want_hit=False, want_accum=False,
           out=None, rows=None, build=None):
    """rt_render into host memory.  Returns dict(rgba, hit, accum, stats)."""
    rgba = out if out is not None else np.zeros((height, width, 4), dtype=np.uint8)
    assert rgba.dtype == np.uint8 and rgba.shape == (height, width, 4) and rgba.flags.c_contiguous
    hit = np.full((height, width), -2, dtype=np.int32) if want_hit else None
    acc = np.zeros((height, width, 3), dtype=np.float32) if want_accum else None
    st = Stats()
    o = _opts(mode, strict_fp, seed, sample_base, chunk_spp, tile_rows, part_index, part_count, rows, build)
    o.hit_index_out = hit.ctypes.data if want_hit else None
    o.accum_out = acc.ctypes.data if want_accum else None
    o.stats = C.addressof(st)
    _check(lib().rt_render(C.byref(cam), rgba.ctypes.data, width, height, spp, max_depth, C.byref(o)),
           "rt_render")
    return dict(rgba=rgba, hit=hit, accum=acc, stats=st)


def render_progressive(cam, width, height, spp, max_depth, frame_index, mode=MODE_MC, strict_fp=False,
                       seed=0, sample_base=0, chunk_spp=0, want_accum=False, out=None, build=None):
    """rt_render_progressive: adds one frame of `spp` samples; returns the image of all frames so far."""
    rgba = out if out is not None else np.zeros((height, width, 4), dtype=np.uint8)
    acc = np.zeros((height, width, 3), dtype=np.float32) if want_accum else None
    st = Stats()
    o = _opts(mode, strict_fp, seed, sample_base, chunk_spp, 0, 0, 0, None, build)
    o.accum_out = acc.ctypes.data if want_accum else None
    o.stats = C.addressof(st)
    _check(lib().rt_render_progressive(C.byref(cam), rgba.ctypes.data, width, height, spp, max_depth,
                                       C.byref(o), frame_index), "rt_render_progressive")
    return dict(rgba=rgba, accum=acc, stats=st)


NOALLOC_SCRATCH_BYTES = 64 << 20      # RT_NOALLOC_SCRATCH_BYTES


def chunk_for(width, height, spp, chunk_spp=CHUNK_AUTO, progressive=False):
    """rt_chunk_for: the samples per work unit a render of this full image size uses."""
    return int(lib().rt_chunk_for(width, height, spp, chunk_spp, int(bool(progressive))))


def render_device(cam, rgba_dev_ptr, width, height, spp, max_depth, mode=MODE_MC, strict_fp=False,
                  seed=0, sample_base=0, chunk_spp=0, tile_rows=0, part_index=0, part_count=0,
                  hit_dev_ptr=None, accum_dev_ptr=None, rows=None, no_alloc=False, build=None):
    """rt_render_device: outputs are raw device pointers (e.g. torch.Tensor.data_ptr()).
    no_alloc=True calls rt_render_device_noalloc (scratch in the library's static device array)."""
    st = Stats()
    o = _opts(mode, strict_fp, seed, sample_base, chunk_spp, tile_rows, part_index, part_count, rows, build)
    o.hit_index_out = hit_dev_ptr
    o.accum_out = accum_dev_ptr
    o.stats = C.addressof(st)
    fn = lib().rt_render_device_noalloc if no_alloc else lib().rt_render_device
    _check(fn(C.byref(cam), rgba_dev_ptr, width, height, spp, max_depth, C.byref(o)),
           "rt_render_device_noalloc" if no_alloc else "rt_render_device")
    return st
