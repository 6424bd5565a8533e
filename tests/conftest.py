import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
# the tests drive the own NT GEMM at shapes far below the size from which the dispatch prefers it over the library
os.environ.setdefault("DPA_GEMM_MIN_TILES", "0")

REFERENCE_ROOT = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this process")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def reference_model_module():
    """The reference single-gpu/model.py imported by path (source only; used as an oracle)."""
    import importlib.util
    path = os.path.join(REFERENCE_ROOT, "single-gpu", "model.py")
    if not os.path.exists(path):
        pytest.skip("reference tree not mounted")
    spec = importlib.util.spec_from_file_location("reference_model", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["reference_model"] = mod
    spec.loader.exec_module(mod)
    return mod
