"""Times the five a-trous passes on a 1080p 4-spp frame of the scene named by argv[1] (default: the metric frame) and on a
cleared image (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import hiplib, workloads
scene = sys.argv[1] if len(sys.argv) > 1 else "scene"
wl = workloads.Workload(scene, 1920, 1080)
ctx = hiplib.Context(0)
ctx.upload_scene(wl.buffers); ctx.resize(1920, 1080); ctx.profile(True)
def timed(label):
    for _ in range(3): ctx.filter(5)
    ctx.sync(); ctx.timings(reset=True)
    for _ in range(20): ctx.filter(5)
    ctx.sync(); t = ctx.timings()
    print(label, "filter ms per frame", round(t.ms[1] / 20, 4))
timed("cleared image")
ctx.trace_samples(wl.rays, wl.eye, spp=4, max_bounces=5, num_emitters=wl.num_emitters)
timed(f"{scene} frame ")
out = ctx.readback(hiplib.IMAGE_FILTERED)
import hashlib; print("sha", hashlib.sha256(out.tobytes()).hexdigest()[:16])
