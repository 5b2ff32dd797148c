"""rt_headless's command line up to the point where it needs a GPU (CPU container: no device).  Errors must be loud —
a message on stderr and a non-zero exit status — never a hang and never a silent fallback."""
import os
import subprocess

import pytest

from raytracer_amd import assets
from tests.util import ROOT

HEADLESS = os.path.join(ROOT, "raytracer_amd", "lib", "rt_headless")


def run(*args, timeout=120):
    assert os.path.exists(HEADLESS), "raytracer_amd/lib/rt_headless is missing (make headless)"
    return subprocess.run([HEADLESS] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout)


def test_bad_arguments_are_refused():
    obj = assets.path("room.obj")
    assert run().returncode != 0
    r = run(obj, "--gpus", 99)
    assert r.returncode != 0 and "bad --gpus" in r.stderr
    r = run(obj, "--kernel", "trace9")
    assert r.returncode != 0 and "unknown kernel" in r.stderr
    r = run(obj, "--keys", "1:NOSUCHKEY", "--frames", 2)
    assert r.returncode != 0 and "unknown key" in r.stderr
    r = run("/nonexistent.obj")
    assert r.returncode != 0 and "NO TRIANGLES LOADED" in r.stderr


def test_gpus_2_without_devices_fails_loudly(tmp_path):
    """Scene, BVH, input script and tiling are prepared on the host; the first thing that needs a device (the RCCL id, then
    one context per GPU) fails: the error names the call, the exit status is non-zero, nothing hangs."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this machine has the devices")
    r = run(assets.path("room.obj"), "--gpus", 2, "--size", "64x48", "--frames", 2, "--filter", 2, "--keys", "1:EQUAL",
            "--out", tmp_path / "frame", timeout=90)
    assert r.returncode != 0, r.stdout
    assert "ERROR::RT::" in r.stderr and ("rt_comm_unique_id" in r.stderr or "NO USABLE GPU" in r.stderr)
