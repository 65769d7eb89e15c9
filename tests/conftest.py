import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
CORNELL = os.path.join(GOLDEN, "cornell_box.gltf")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_l2(a, b):
    """Relative L2 distance of two radiance images (rgb only)."""
    a = np.asarray(a, dtype=np.float64)[..., :3]
    b = np.asarray(b, dtype=np.float64)[..., :3]
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))


@pytest.fixture(scope="session")
def cornell():
    import logipathtracer_amd as L
    return L.Scene(CORNELL)


@pytest.fixture(scope="session")
def cornell_oracle(cornell):
    from oracle import binding as O
    return O.OracleScene(cornell.objects, cornell.object_bvh, cornell.vertices, cornell.mesh_bvh)


@pytest.fixture(scope="session")
def device():
    import logipathtracer_amd as L
    dev = L.Device(0)
    yield dev
    dev.close()
