import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mm():
    """The product package (markovmodels.jl_amd); building it first if the
    library is missing (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as ge

    if not os.path.exists(os.path.join(ge.PKG_DIR, "libmarkovmodels_amd.so")):
        ge.build()
    return ge.load_package()


@pytest.fixture(scope="session")
def oracle():
    import __graft_entry__ as ge

    return ge.load_oracle()


@pytest.fixture(scope="session")
def wl(mm):
    import importlib

    return importlib.import_module(mm.__name__ + ".workloads")
