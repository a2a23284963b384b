"""rt_render_device_noalloc: the entry point that allocates nothing (its per-chunk sums and counters
live in a fixed device array of the library) against rt_render_device, through the C ABI: two calls
in a row, the second with a smaller input, outputs bit for bit equal; the size limit is an error,
not a fault; and the device's free memory does not move across the calls."""
import numpy as np
import pytest
import torch  # before the library: PyTorch's bundled HIP runtime must load first

from conftest import load_scene

pytestmark = pytest.mark.gpu


def _both(rt, torch, cam, W, H, spp, depth, **kw):
    rows = rt.part_rows(H, kw.get("tile_rows", 0), kw.get("part_index", 0), kw.get("part_count", 0))
    out = []
    for no_alloc in (False, True):
        rgba = torch.zeros((rows, W, 4), dtype=torch.uint8, device="cuda:0")
        acc = torch.zeros((rows, W, 3), dtype=torch.float32, device="cuda:0")
        hit = torch.full((rows, W), -7, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        st = rt.render_device(cam, rgba.data_ptr(), W, H, spp, depth, hit_dev_ptr=hit.data_ptr(),
                              accum_dev_ptr=acc.data_ptr(), no_alloc=no_alloc, **kw)
        torch.cuda.synchronize()
        out.append((rgba.cpu().numpy(), acc.cpu().numpy(), hit.cpu().numpy(), int(st.rays), int(st.samples), int(st.chunk_spp)))
    return out


def _same(a, b):
    assert np.array_equal(a[0], b[0])
    assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    assert np.array_equal(a[2], b[2])
    assert a[3:] == b[3:] and a[3] > 0


@pytest.mark.parametrize("build", ["exact", "strict", "fast"])
def test_no_alloc_entry_point_matches_the_allocating_one(rt, build):
    sph = load_scene("sf")
    rt.set_world(sph)
    # first call: 640x360, three chunks per pixel (11 MB of per-chunk sums in the scratch array)
    W, H = 640, rt.height_for(640)
    cam = rt.camera(W, H)
    ref, got = _both(rt, torch, cam, W, H, 6, 50, build=build, seed=5, chunk_spp=2)
    _same(ref, got)
    assert (got[0][..., 3] == 255).all() and (got[2] >= -1).all()
    # second call, smaller input: 200x112, one part of three, other seed; the array keeps the first
    # call's sums beyond what this call writes, and its counters must start from zero again
    free0 = torch.cuda.mem_get_info()[0]
    W2, H2 = 200, rt.height_for(200)
    cam2 = rt.camera(W2, H2)
    ref2, got2 = _both(rt, torch, cam2, W2, H2, 5, 20, build=build, seed=9, chunk_spp=0,
                       tile_rows=8, part_index=1, part_count=3)
    _same(ref2, got2)
    # and the first again: nothing of the smaller call is left over
    ref3, got3 = _both(rt, torch, cam, W, H, 6, 50, build=build, seed=5, chunk_spp=2)
    _same(ref, got3)
    # constant-u mode and depth 0 take the same path
    _same(*_both(rt, torch, cam2, W2, H2, 1, 50, build=build, mode=rt.MODE_CONST_U, chunk_spp=0))
    a, b = _both(rt, torch, cam2, W2, H2, 2, 0, build=build, chunk_spp=0)
    assert np.array_equal(a[0], b[0]) and (b[0][..., :3] == 0).all()
    free1 = torch.cuda.mem_get_info()[0]
    assert free1 >= free0            # torch's own caching allocator served the test's tensors from its pool


def test_no_alloc_entry_point_rejects_what_does_not_fit(rt):
    rt.set_world(load_scene("s3"))
    W, H = 1920, rt.height_for(1920)
    cam = rt.camera(W, H)
    rgba = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda:0")
    # 1920x1080 x 16 B = 33 MB per chunk: two chunks fit the 64 MiB array, three do not
    st = rt.render_device(cam, rgba.data_ptr(), W, H, 4, 5, chunk_spp=2, no_alloc=True)
    assert int(st.samples) == W * H * 4
    torch.cuda.synchronize()
    assert (rgba[..., 3] == 255).all().item()
    with pytest.raises(rt.RtError, match="scratch array holds"):
        rt.render_device(cam, rgba.data_ptr(), W, H, 6, 5, chunk_spp=2, no_alloc=True)
    with pytest.raises(rt.RtError, match="rows is not supported"):
        rt.render_device(cam, rgba.data_ptr(), W, H, 1, 5, chunk_spp=0, rows=[0, 1], no_alloc=True)
    # the library is still usable and the allocating entry takes the same call
    rt.render_device(cam, rgba.data_ptr(), W, H, 6, 5, chunk_spp=2)
