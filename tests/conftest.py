import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The fp64 oracle is many small ATen calls.  A GPU box gives a one-GPU job a SHARE of a large host (16 cores of 128+): torch's default
    # intra-op pool, sized by the host's core count, then spins far more threads than the job may run and the oracle is ~30 x slower than on
    # 8 real cores (round 5: 26 s against 0.8 s per B = 4 iteration; the GPU suite took 866 s, 650 of them here).  Size the pool by what
    # the process can use.
    import torch
    from bench import host_cores
    torch.set_num_threads(host_cores(cap=16))


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("gesture-generation-from-trimodal-context_amd")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
