"""Development aid: render a few frames of one workload (for rocprofv3 --kernel-trace; see scripts/stage_times.sh).
usage: python scripts/stage_frames.py scene|room [frames] [spp] [bounces] [filter]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import hiplib, workloads
name = sys.argv[1] if len(sys.argv) > 1 else "scene"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 4
mb = int(sys.argv[4]) if len(sys.argv) > 4 else 5
it = int(sys.argv[5]) if len(sys.argv) > 5 else 5
W, H = 1920, 1080
wl = workloads.Workload(name, W, H)
ctx = hiplib.Context(0)
ctx.upload_scene(wl.buffers)
ctx.resize(W, H)
for f in range(frames):
    ctx.trace_samples(wl.rays, wl.eye, spp=spp, max_bounces=mb, num_emitters=wl.num_emitters)
    if it:
        ctx.filter(it)
    ctx.sync()
print("walk_stop", ctx.walk_stop())
ctx.close()
