import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _guarded_allocations(request):
    """GROVE_TEST_GUARD=1: every GPU test runs with the out-of-bounds / unwritten-read screen of tests/test_guards_gpu.py around it —
    each torch.empty / torch.zeros CUDA tensor carved out of a patterned buffer (margins checked when the test ends), fresh tensors
    NaN-poisoned. Off by default: the margins are held until the test ends (memory), full-width cases would not fit.
    `GROVE_TEST_GUARD=1 python -m pytest tests/test_branches_gpu.py tests/test_model_gpu.py -m gpu -k "not full_width"`"""
    if os.environ.get("GROVE_TEST_GUARD") != "1" or request.node.get_closest_marker("gpu") is None or "test_guards_gpu" in request.node.nodeid:
        yield
        return
    from test_guards_gpu import GuardedAllocs
    with GuardedAllocs(poison=os.environ.get("GROVE_TEST_GUARD_POISON", "1") == "1") as ga:
        yield
    ga.check(request.node.nodeid)
