import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must not silently pass: skip with a reason instead of erroring at import
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _seed_per_test(request):
    """Tests that draw device tensors from torch's global generators get the same data whatever ran before them (the seed is a
    function of the test's id): a premise like "the two arithmetic paths are distinguishable on this input" must not depend on the
    order or the selection of the run."""
    try:
        import torch
        import zlib
        torch.manual_seed(zlib.crc32(request.node.nodeid.encode()))
    except ImportError:
        pass
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR


def golden_cases():
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and not f.startswith("llmc_"))
