"""Development aid (needs librt_hip.so built with `make hip EXTRA_HIPFLAGS=-DRT_WAVE_TRACE`): per-wave timeline of the
extend kernel for one frame.  usage: python scripts/wave_trace.py [scene|room] [spp]"""
import ctypes, sys
sys.path.insert(0, '.')
import numpy as np
import os
from raytracer_amd import hiplib, workloads
hiplib.LIB_PATH = hiplib.LIB_PATH.replace("librt_hip.so", "librt_hip_trace.so")
name = sys.argv[1] if len(sys.argv) > 1 else "scene"
wl = workloads.Workload(name, 1920, 1080)
ctx = hiplib.Context(0); ctx.upload_scene(wl.buffers); ctx.resize(1920, 1080)
for rep in range(3):
    ctx.trace_samples(wl.rays, wl.eye, spp=4, max_bounces=5, num_emitters=wl.num_emitters); ctx.sync()
buf = np.zeros((8, 8192, 16), np.int64)
lib = ctypes.CDLL(hiplib.LIB_PATH)
lib.rt_debug_wave_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = lib.rt_debug_wave_trace(buf.ctypes.data, buf.nbytes)
print("rc", rc)
for st in range(8):
    a = buf[st]
    a = a[a[:, 1] > 0]
    if not len(a):
        continue
    t0, t1 = a[:, 0], a[:, 1]
    base = t0.min()
    dur, end, start = (t1 - t0) / 100.0, (t1 - base) / 100.0, (t0 - base) / 100.0
    print(f"stage {st}: waves {len(a)} rpw {a[0, 7]} rays {a[:, 2].sum()} launch {end.max():.1f} us | start max {start.max():.1f} | wave dur mean {dur.mean():.1f} p50 {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} p99 {np.percentile(dur, 99):.1f} max {dur.max():.1f}")
    print(f"    rays/wave mean {a[:, 2].mean():.1f} max {a[:, 2].max()} | outer mean {a[:, 3].mean():.1f} max {a[:, 3].max()} | coop rounds mean {a[:, 4].mean():.1f} max {a[:, 4].max()} | pairs mean {a[:, 5].mean():.0f} max {a[:, 5].max()} | walk mean {a[:, 6].mean():.0f} max {a[:, 6].max()}")
    for i in np.argsort(-dur)[:4]:
        print(f"      slow wave {i}: {start[i]:.1f}->{end[i]:.1f} us rays {a[i, 2]} outer {a[i, 3]} coop {a[i, 4]} pairs {a[i, 5]} walk {a[i, 6]}")
    tot = a[:, 8].astype(float)
    names = ["refill", "walk", "setup", "rounds", "finish", "tail"]
    slow = np.argsort(-dur)[:64]
    print("    cycles/wave (all): total %.0f | " % tot.mean() + " ".join(f"{n} {a[:, 9 + i].mean():.0f}" for i, n in enumerate(names)) + f" | retests {a[:, 15].mean():.1f}")
    print("    cycles/wave (64 slowest): total %.0f | " % tot[slow].mean() + " ".join(f"{n} {a[slow, 9 + i].mean():.0f}" for i, n in enumerate(names)) + f" | outer {a[slow, 3].mean():.1f} coop {a[slow, 4].mean():.1f} pairs {a[slow, 5].mean():.0f} walk {a[slow, 6].mean():.0f} retests {a[slow, 15].mean():.1f}")
    print(f"    end-time percentiles p10 {np.percentile(end, 10):.1f} p50 {np.percentile(end, 50):.1f} p90 {np.percentile(end, 90):.1f} p99 {np.percentile(end, 99):.1f}")
