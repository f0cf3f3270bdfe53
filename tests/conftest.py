import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
ALL_SHAPES = ["1d1r", "1d2r", "star2d1r", "box2d1r", "star2d3r", "box2d3r", "star3d1r", "box3d1r"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Build what is missing BEFORE collection: the engine, the oracle and -- where /root/reference is present -- the
    # reference's own test_cpu (oracle/_ref), whose availability some tests check in skipif() at import time.
    import __graft_entry__ as g

    g.build(only_if_missing=True)


def load_golden(shape):
    z = np.load(os.path.join(GOLDEN, f"{shape}.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def engine_built():
    """Build the HIP engine (cross-compiles without a GPU) once per session if it is missing."""
    import __graft_entry__ as g

    g.build(only_if_missing=True)
    return True


def has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False
