import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_weights():
    return dict(np.load(os.path.join(GOLDEN, "student_lambda_00_weights.npz")))


@pytest.fixture(scope="session")
def golden_io():
    z = np.load(os.path.join(GOLDEN, "student_lambda_00_io.npz"))
    return {k: z[k] for k in z.files}


CHARACTERS = ["lambda_00", "lambda_01"]      # both students the reference ships (data/character_models/)


@pytest.fixture(scope="session")
def char_weights():
    return {c: dict(np.load(os.path.join(GOLDEN, f"student_{c}_weights.npz"))) for c in CHARACTERS}


@pytest.fixture(scope="session")
def char_io():
    out = {}
    for c in CHARACTERS:
        z = np.load(os.path.join(GOLDEN, f"student_{c}_io.npz"))
        out[c] = {k: z[k] for k in z.files}
    return out


@pytest.fixture(scope="session")
def built():
    """Make sure the native pieces exist (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    from tha4_amd import _build
    _build.build_native()
    g.build_emulator()
    return True
