import os
import sys

import pytest

# PyTorch-ROCm bundles its own HIP runtime and RCCL (same sonames as /opt/rocm's).  Some tests use torch.cuda next to the
# launcher; for torch.cuda to be usable in such a process torch's copies must be the ones loaded, i.e. torch is imported
# FIRST ("PyTorch in the same process").  Either order exits cleanly since round 3 (the launcher opens RCCL
# RTLD_LOCAL after probing for a loaded copy: tests/test_rccl_import_order.py runs both orders in child processes); in
# the other order torch.cuda merely reports no device.
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


_WALK_DEBUG_EXPORTS = ("rt_debug_walk_violations_per_pixel", "rt_debug_walk_violations_rl",
                       "rt_debug_walk_violations_wavefront")


@pytest.fixture(autouse=True)
def _no_walk_violations(request):
    """With a -DRT_DEBUG_WALK build of the launcher (make debugwalk; RT_HIP_LIB=raytracer_amd/lib/variants/
    librt_hip_debugwalk.so) every link and leaf range a kernel follows is checked on the device; after each GPU test
    the violation counters of every translation unit must be zero.  With the ordinary build this does nothing."""
    yield
    if "gpu" not in request.keywords:
        return
    import ctypes
    from raytracer_amd import hiplib
    L = hiplib._lib
    if L is None or not hasattr(L, _WALK_DEBUG_EXPORTS[0]):
        return
    for name in _WALK_DEBUG_EXPORTS:
        out = (ctypes.c_int * 4)()
        assert getattr(L, name)(out) == 0, name
        assert list(out) == [0, 0, 0, 0], (name, "link out of range / leaf beyond the triangle stream / packed loop on "
                                                 "an unpacked scene", list(out))
