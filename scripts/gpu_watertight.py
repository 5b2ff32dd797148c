"""Cost of RT_FLAG_WATERTIGHT (opt-in triangle test) against the default on the bench scenes."""
import sys, time
sys.path.insert(0, '.')
from raytracer_amd import hiplib, workloads
ctx = hiplib.Context(0)
for name in ("scene", "room"):
    wl = workloads.Workload(name, 1920, 1080)
    ctx.upload_scene(wl.buffers); ctx.resize(1920, 1080)
    for label, flag in (("reference test", 0), ("watertight", hiplib.FLAG_WATERTIGHT)):
        for rep in range(8):
            if rep == 4:
                ctx.sync(); t = time.perf_counter()
            ctx.trace_samples(wl.rays, wl.eye, spp=4, max_bounces=5, flags=flag, num_emitters=wl.num_emitters)
        ctx.sync()
        print(f"{name:6s} {label:15s} {(time.perf_counter() - t) * 250:.3f} ms/frame (trace only)")
