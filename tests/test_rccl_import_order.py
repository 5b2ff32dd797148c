"""rt_comm.hip opens RCCL at run time.  Opened RTLD_GLOBAL (round 2) it made a PyTorch imported AFTERWARDS resolve nccl*
against that copy next to its own bundled HIP runtime, and the interpreter aborted at exit ("double free or corruption").
It is opened RTLD_LOCAL now, after an RTLD_NOLOAD probe for a copy the process already holds: both import orders must
end in a clean exit.  Runs in child processes; without a GPU ncclGetUniqueId itself fails (RT_EHIP), which is fine here —
the exit status is what is checked; on the GPU box the id must also be made."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROLOGUE = "import ctypes, sys\nsys.path.insert(0, %r)\n" % ROOT
RCCL = ("from raytracer_amd import hiplib\nL = hiplib.lib()\nbuf = (ctypes.c_char * 128)()\n"
        "print('unique_id rc', L.rt_comm_unique_id(buf, 128), flush=True)\n")
TORCH = "import torch\nprint('torch ok', torch.cuda.is_available(), flush=True)\n"


def run(body):
    return subprocess.run([sys.executable, "-c", PROLOGUE + body], capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("order", ["rccl_then_torch", "torch_then_rccl"])
def test_clean_exit_in_both_import_orders(order):
    r = run(RCCL + TORCH if order == "rccl_then_torch" else TORCH + RCCL)
    assert r.returncode == 0, (order, r.returncode, r.stdout[-500:], r.stderr[-2000:])
    assert "unique_id rc" in r.stdout and "torch ok" in r.stdout
    assert "double free" not in r.stderr and "corruption" not in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("order", ["rccl_then_torch", "torch_then_rccl"])
def test_id_is_made_and_exit_is_clean_on_the_gpu_box(order):
    r = run(RCCL + TORCH if order == "rccl_then_torch" else TORCH + RCCL)
    assert r.returncode == 0, (order, r.returncode, r.stdout[-500:], r.stderr[-2000:])
    assert "unique_id rc 0" in r.stdout and "torch ok" in r.stdout
    # torch imported first: ONE HIP runtime and ONE RCCL in the process (the launcher's DT_NEEDED libamdhip64.so.7 and
    # its RTLD_NOLOAD probe both resolve to torch's bundled copies), and torch sees the GPU.  The launcher first: it
    # has bound /opt/rocm's runtime, torch then loads its own second copy by path, which finds no device — torch's GPU
    # side is unusable in that order ("PyTorch in the same process"), but nothing aborts.
    if order == "torch_then_rccl":
        assert "torch ok True" in r.stdout
