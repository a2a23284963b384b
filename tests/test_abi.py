"""The C-ABI library: loads, exports every symbol include/rt.h declares, and its host-only
entry points (no GPU needed) behave as documented."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_camera_golden
import raytracer_amd as rt


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(rt.EXPORTS)


def test_library_exports_every_declared_symbol():
    L = C.CDLL(rt.LIB_PATH)
    for name in declared_symbols():
        assert getattr(L, name) is not None


def test_struct_sizes_match_reference_layout():
    assert rt.SPHERE_DTYPE.itemsize == 40         # sphere_t, Main.c:74-79
    assert C.sizeof(rt.Camera) == 76
    assert C.sizeof(rt.CameraParams) == 48


@pytest.mark.parametrize("width", [100, 400, 1200, 1920, 3840])
def test_camera_setup_matches_reference(width):
    arr, W, H = load_camera_golden(width)
    assert rt.height_for(width) == H
    cam = rt.camera(W, H)
    assert np.array_equal(rt.camera_as_array(cam).view("u4"), arr.view("u4"))


def test_part_rows_partition_the_image():
    for height in (1, 15, 16, 17, 225, 2160):
        for tile_rows in (1, 8, 16):
            for count in (1, 2, 3, 8):
                rows = [rt.part_row_map(height, tile_rows, i, count) for i in range(count)]
                allr = np.sort(np.concatenate(rows))
                assert np.array_equal(allr, np.arange(height))
                for r in rows:
                    assert np.all(np.diff(r) > 0)


def test_render_before_init_is_an_error_not_a_crash():
    rt.shutdown()
    cam = rt.camera(100)
    buf = np.zeros((56, 100, 4), np.uint8)
    with pytest.raises(rt.RtError, match="rt_init"):
        rt.render(cam, 100, 56, 1, 1, out=buf)
    with pytest.raises(rt.RtError, match="rt_init"):
        rt.set_world(np.zeros(1, rt.SPHERE_DTYPE))


def test_scene_generator_reproduces_the_reference_scenes():
    """rt_scene_generate against the sphere lists the compiled reference produced."""
    import conftest
    gold_sf = np.fromfile(os.path.join(conftest.GOLD, "scene_sf.bin"), dtype=rt.SPHERE_DTYPE)
    gold_s3 = np.fromfile(os.path.join(conftest.GOLD, "scene_s3.bin"), dtype=rt.SPHERE_DTYPE)
    assert rt.scene_generate(11, 0).tobytes() == gold_sf.tobytes()
    assert rt.scene_generate(0, 0).tobytes() == gold_s3.tobytes()
    other = rt.scene_generate(11, 1)
    assert len(other) > 400 and other.tobytes() != gold_sf.tobytes()
    big = rt.scene_generate(20, 0)
    assert 1500 < len(big) <= 1604 and np.all(big["radius"][1:-3] == np.float32(0.2))
    n = C.c_size_t(0)
    buf = np.zeros(10, dtype=rt.SPHERE_DTYPE)
    assert rt.lib().rt_scene_generate(11, 0, buf.ctypes.data, 10, C.byref(n)) == -1 and n.value == 488
