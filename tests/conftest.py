import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        from expressionmatrix2_amd import capi
        return capi.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; when selected on a box without a GPU they must FAIL, not skip,
    # so that a silent fallback can never look green.
    pass


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    return oracle_binding.load_oracle()


@pytest.fixture(scope="session")
def hostchecks():
    import oracle_binding
    return oracle_binding.load_host_checks()


@pytest.fixture(scope="session")
def reflib():
    import oracle_binding
    lib = oracle_binding.load_ref()
    if lib is None:
        pytest.skip("oracle/_ref/libem2ref.so not built (needs /root/reference)")
    return lib


@pytest.fixture(scope="session")
def reflayout():
    import oracle_binding
    lib = oracle_binding.load_ref_layout()
    if lib is None:
        pytest.skip("oracle/_ref/libem2ref_layout.so not built (needs /root/reference)")
    return lib
