import sys
sys.path.insert(0, '.')
import numpy as np
from raytracer_amd import hiplib, hostlib
from tests.test_gpu_watertight import _Mesh
ctx = hiplib.Context(0)
for n in (256, 1024):
    wl = _Mesh(n, 3840, 2160)
    ctx.upload_scene(wl.buffers); ctx.resize(wl.width, wl.height)
    tot = {0: 0, hiplib.FLAG_WATERTIGHT: 0}
    for k in range(12):
        eye = np.array([0.1 + 0.37 * k, 2.0 + 0.11 * k, 0.2 - 0.29 * k], np.float32)
        rays = hostlib.camera_rays(eye=tuple(eye), look_at=(0.3 * k, 0.0, -0.2 * k), up=(0.0, 0.0, 1.0), fov=60.0, width=3840, height=2160)
        for flag in tot:
            ctx.trace_samples(rays, eye, spp=1, max_bounces=0, flags=hiplib.FLAG_AUX | flag, num_emitters=0)
            tri = ctx.readback(hiplib.IMAGE_TRIDX)
            # holes = misses whose 4-neighbours all hit
            miss = tri < 0
            inner = miss[1:-1, 1:-1] & ~miss[:-2, 1:-1] & ~miss[2:, 1:-1] & ~miss[1:-1, :-2] & ~miss[1:-1, 2:]
            tot[flag] += int(inner.sum())
    print(n, "isolated misses over 12 views: reference", tot[0], "watertight", tot[hiplib.FLAG_WATERTIGHT])
