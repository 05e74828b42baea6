import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # every table export in the tests recounts the keys on the host and fails if the device's own statistic differs (the product path
    # trusts the statistic: bfcg_export_table)
    os.environ.setdefault("BFC_GPU_RECOUNT", "1")


@pytest.fixture(scope="session")
def g1():
    from bfc_amd import gen
    rs = gen.fixture("g1")
    return rs, rs.reads()


@pytest.fixture(scope="session")
def g42():
    from bfc_amd import gen
    rs = gen.fixture("g42")
    return rs, rs.reads()


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library; GPU tests call through its C ABI only."""
    import bfc_amd
    from bfc_amd import build, _lib
    if not os.path.exists(_lib.SO):
        build.build()
    return bfc_amd
