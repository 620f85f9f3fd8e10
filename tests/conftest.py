import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    # the parity tests run the fp32 oracle live on the GPU box's host: every logical CPU oversubscribes it (83.9 s per step on 128
    # threads against 53.9 s on 6 for the same UNet) — keep to a few dozen threads
    torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
    from storygen_amd import ops
    arch = ops.device_arch()
    assert arch == 950, f"libstorygen_hip is built for gfx950, device reports gfx{arch}"
    return torch.device("cuda:0")


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
