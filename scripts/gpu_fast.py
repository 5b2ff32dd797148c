"""Development aid: parity mode vs RT_FLAG_FAST_SHADOWS on the bench workloads."""
import sys
sys.path.insert(0, '.')
from scripts.gpu_perf import run
from raytracer_amd import hiplib
for name in ("scene", "room"):
    run(name, 1920, 1080)
    run(name, 1920, 1080, flags=hiplib.FLAG_FAST_SHADOWS)
