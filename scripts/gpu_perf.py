"""Development aid: kernel-level timing of the trace + filter path on a few workloads."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from raytracer_amd import hiplib, workloads

def run(name, W, H, spp=4, mb=5, iters=5, reps=5, **kw):
    wl = workloads.Workload(name, W, H)
    ctx = hiplib.Context(0)
    ctx.upload_scene(wl.buffers); ctx.resize(W, H)
    ctx.trace_samples(wl.rays, wl.eye, spp=spp, max_bounces=mb, **kw); ctx.filter(iters); ctx.sync()
    ctx.reset_counters(); ctx.profile(True); ctx.timings(reset=True)
    t = time.perf_counter()
    for _ in range(reps):
        ctx.trace_samples(wl.rays, wl.eye, spp=spp, max_bounces=mb, **kw)
        ctx.filter(iters)
    ctx.sync(); wall = (time.perf_counter() - t) / reps
    c, tm = ctx.counters(), ctx.timings()
    tr = tm.trace_ms / reps
    print(f"{str(kw):28s} {name:8s} {W}x{H} spp{spp} mb{mb}: frame wall {wall*1e3:8.3f} ms | trace {tr:8.3f} ms ({tr/spp:7.3f}/launch) filter {tm.filter_ms/reps:6.3f} ms | "
          f"rays/frame {c.rays//reps:>10d} visits/ray {c.node_visits/max(c.rays,1):6.1f} tests/ray {c.tri_tests/max(c.rays,1):6.1f} | {c.rays/reps/tr/1e3:8.1f} Mrays/s (trace only)", flush=True)
    print("      per-kernel ms/frame:", {k: round(v[0]/reps, 3) for k, v in tm.as_dict().items()}, flush=True)
    ctx.close()

if __name__ == "__main__":
    PP = dict(batch=1, flags=hiplib.FLAG_PER_PIXEL_KERNEL)
    for kw in (dict(), dict(batch=1), PP):
        run('scene', 1920, 1080, **kw)
        run('room', 1920, 1080, **kw)
    run('suzanne', 1920, 1080)
    run('teapot', 640, 480, spp=1, iters=0)
    run('scene', 3840, 2160, spp=4)
