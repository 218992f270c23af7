import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built_tree():
    """The C-ABI library, the JVM-less driver and the oracle library are build products (git-ignored); a fresh checkout gets them from
    __graft_entry__.build() -- hipcc cross-compiles gfx950 without a GPU, about two minutes -- before the first test needs them."""
    need = [os.path.join(ROOT, "carskit_amd", "lib", "libcarskit_mi355x.so"), os.path.join(ROOT, "carskit_amd", "bin", "carskit-mi355x"),
            os.path.join(ROOT, "oracle", "libcarskit_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()
    yield
