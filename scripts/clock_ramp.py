"""Does a fresh box run the first seconds of GPU work slower than the rest?  Bursts of 20 metric frames (as bench.py's
timed region: queued back to back, synchronised at both ends), one after the other from the first moment the context
exists, each with the time since the first burst began.  Run it as the FIRST GPU command of a gpurun call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from raytracer_amd import hiplib, workloads  # noqa: E402

W, H = 1920, 1080
wl = workloads.Workload("scene", W, H)
ctx = hiplib.Context(0)
ctx.upload_scene(wl.buffers)
ctx.resize(W, H)


def step():
    ctx.trace_samples(wl.rays, wl.eye, spp=4, max_bounces=5, flags=0, num_emitters=wl.num_emitters)
    ctx.filter(5)


gap = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0   # idle seconds between bursts (does the GPU fall asleep again?)
start = time.perf_counter()
for burst in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    ctx.sync()
    t1 = time.perf_counter()
    print(f"burst {burst:3d} at {t0 - start:7.3f} s: {1e3 * (t1 - t0) / 20:.4f} ms per frame (walk_stop {ctx.walk_stop()})", flush=True)
    if gap:
        time.sleep(gap)
ctx.close()
