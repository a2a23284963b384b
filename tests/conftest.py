import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a gfx950 GPU (run on the MI355X box)")


def load_scene(name):
    from oracle import refcpu
    return np.fromfile(os.path.join(GOLD, "scene_%s.bin" % name), dtype=refcpu.SPHERE_DTYPE)


def load_camera_golden(width):
    z = np.load(os.path.join(GOLD, "cameras.npz"))
    wh = z["wh_%d" % width]
    return z["cam_%d" % width], int(wh[0]), int(wh[1])


def golden(name):
    return np.load(os.path.join(GOLD, name))


def psnr(a, b):
    """PSNR in dB between two RGBA8 images (RGB channels only)."""
    a = a[..., :3].astype(np.float64)
    b = b[..., :3].astype(np.float64)
    mse = np.mean((a - b) ** 2)
    return 99.0 if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)


@pytest.fixture(scope="session")
def rt():
    """The product library, initialised on GPU 0.  No fallback: raises if unavailable."""
    try:
        import torch  # noqa: F401  (PyTorch's bundled HIP runtime must load before the system one the library links)
    except ImportError:
        pass
    import raytracer_amd as rt_
    rt_.init(0)
    yield rt_
    rt_.shutdown()
