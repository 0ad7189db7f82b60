import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _has_gpu() -> bool:
    try:
        from raytracer_amd import abi
        return abi.load_library().rt_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; without a device they are skipped, not failed.
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def synthetic_texture(width: int = 1920, height: int = 1080) -> np.ndarray:
    """Deterministic RGBA8 test image of the wall texture's size (the reference's
    wall_texture.jpg is an asset of the reference and is not shipped with the tests)."""
    y, x = np.mgrid[0:height, 0:width].astype(np.uint32)
    r = (x * 7 + y * 3) & 0xFF
    g = ((x >> 2) ^ (y >> 1)) & 0xFF
    b = (x * y + 91) & 0xFF
    a = np.full_like(r, 255)
    return np.stack([r, g, b, a], axis=-1).astype(np.uint8)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def wall_texture():
    return synthetic_texture()
