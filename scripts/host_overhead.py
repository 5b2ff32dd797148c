"""How long does the HOST take to queue one metric frame (rt_trace_batch of 4 samples + rt_filter of 5 passes)?
Compares the Python path bench.py uses (hiplib.Context.trace_samples: builds four rt_trace_params per call) with the
same two C-ABI calls on a parameter array built once."""
import ctypes
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
L = hiplib.lib()


def py_step():
    ctx.trace_samples(wl.rays, wl.eye, spp=4, max_bounces=5, flags=0, num_emitters=wl.num_emitters)
    ctx.filter(5)


plist = []
for k in range(4):
    import numpy as np
    t = np.float32(1.25) + np.float32(0.0625) * np.float32(k)
    bf = np.float32(k) / (np.float32(k) + np.float32(1.0))
    plist.append(hiplib.make_params(wl.rays, wl.eye, t, bf, 5, wl.num_emitters, 0, frame_index=k))
arr = (hiplib.TraceParams * 4)(*plist)
h = ctx._h


def c_step():
    L.rt_trace_batch(h, arr, 4, None)
    L.rt_filter(h, 5, 3.3, 1e-2, 5.5, None)


for name, step in (("python trace_samples", py_step), ("prebuilt params", c_step)):
    for _ in range(12):
        step()
    ctx.sync()
    for n in (20, 100):
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        t1 = time.perf_counter()
        ctx.sync()
        t2 = time.perf_counter()
        print(f"{name:22s} {n:4d} frames: queued in {1e3 * (t1 - t0) / n:.3f} ms per frame, finished {1e3 * (t2 - t0) / n:.3f} ms per frame")
