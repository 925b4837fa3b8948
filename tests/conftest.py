import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("FNX_TEST_MIOPEN", "0") != "1":
        # The reference-side expressions in the tests (utils.loss_utils.ssim = F.conv2d + autograd) go through ATen's
        # own convolution kernels: MIOpen's cold-cache find/compile step aborts the process on a fresh box when it
        # first runs late in a long session (tools/flaky_probe.sh; nothing of this repository is on that stack).  The
        # product path has no conv2d.
        import torch
        torch.backends.cudnn.enabled = False


@pytest.fixture(scope="session")
def oracle():
    from oracle import raster_oracle
    raster_oracle.build()
    return raster_oracle
