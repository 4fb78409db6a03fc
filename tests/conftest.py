import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle runs inside the GPU tests as the checker: on the GPU box's 256-core host torch's default (one thread per core)
    # is 3-4x SLOWER for its GEMMs than 32 threads (measured: 2048 x 4096 x 14336 fp32 at 32 / 64 / 128 / 256 threads = 1.53 / 1.26 /
    # 0.87 / 0.42 TFLOP/s, profiles/r3_host_threads.log)
    try:
        import torch
        torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    except Exception:                                          # pragma: no cover
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
