import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU suite shares an 8-core VM with its own two-rank gloo children and whatever the host schedules: with every core claimed by
    # one process's OpenMP team a descheduled worker stalls each parallel region (a 20-second suite has been seen to take ten minutes)
    import torch
    if not torch.cuda.is_available():
        torch.set_num_threads(max(1, min(4, (os.cpu_count() or 4) // 2)))


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "main16_golden.npz"))
