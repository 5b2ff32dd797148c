"""Experiment: how much would two frames in flight gain?  Two contexts on one GPU render alternate frames
(independent images, streams and pipeline buffers); compared with one context rendering every frame."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import hiplib, workloads
name = sys.argv[1] if len(sys.argv) > 1 else "scene"
wl = workloads.Workload(name, 1920, 1080)
NCTX = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctxs = [hiplib.Context(0) for _ in range(NCTX)]
for c in ctxs:
    c.upload_scene(wl.buffers); c.resize(1920, 1080)
def frame(c):
    c.trace_samples(wl.rays, wl.eye, spp=4, max_bounces=5, num_emitters=wl.num_emitters)
    c.filter(5)
for c in ctxs:
    for _ in range(8): frame(c)
    c.sync()
for label, use in [("one context ", [ctxs[0]])] + [(f"{m} contexts", ctxs[:m]) for m in range(2, NCTX + 1)] * 2:
    n = 40
    t0 = time.perf_counter()
    for k in range(n): frame(use[k % len(use)])
    for c in use: c.sync()
    print(label, round((time.perf_counter() - t0) * 1e3 / n, 4), "ms per frame")
