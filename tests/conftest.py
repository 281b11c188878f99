import os
import sys

import pytest
import torch  # noqa: F401  (loaded before libmpcx.so brings in its own HIP runtime: the other order leaves torch without a GPU)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One mpcx context on device 0, through the C ABI.  Fails (does not skip) when the GPU library cannot run."""
    from mpc_rs_amd import ffi
    c = ffi.Context(0)
    yield c
    c.close()
