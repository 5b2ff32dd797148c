import os
import sys

import pytest

# PyTorch-ROCm bundles its own HIP runtime and RCCL (same sonames as /opt/rocm's).  Some tests use torch next to the
# launcher; the process must then load torch's copies FIRST ("PyTorch in the same process"): loaded
# after librt_hip.so / librccl.so.1 of /opt/rocm, the two sets of libraries end in a double free at interpreter exit.
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs /root/reference and Mesa llvmpipe (build container only)")


def pytest_collection_modifyitems(config, items):
    from oracle import refrun
    if refrun.available():
        return
    skip = pytest.mark.skip(reason="reference oracle unavailable (no /root/reference or oracle/_ref)")
    for item in items:
        if "ref" in item.keywords:
            item.add_marker(skip)


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
