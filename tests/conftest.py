import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("f2-nerf_amd")


@pytest.fixture(scope="session")
def capi(pkg):
    return pkg.capi


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: the HIP path has no CPU fallback")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _default_kernel_routes():
    """Every test starts and ends on the production kernel routes (f2n_set_option is process-wide)."""
    yield
    capi = importlib.import_module("f2-nerf_amd").capi
    if capi._lib is not None:
        for name in capi.option_keys():
            capi.set_option(name, 0)
