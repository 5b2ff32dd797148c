import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from raytracer_amd import hiplib, workloads
W, H = 1920, 1080
wl = workloads.Workload("scene", W, H)
ctx = hiplib.Context(0)
ctx.upload_scene(wl.buffers); ctx.resize(W, H)
ctx.trace_samples(wl.rays, wl.eye, spp=4, max_bounces=5, num_emitters=wl.num_emitters)
ctx.filter(5)
fl = ctx.flat_cells(0); lv = ctx.flat_cells(1)
print("flat cells", fl.mean())
for i in range(5):
    ok = lv > i
    print("pass", i, "cells with result", round(ok.mean(), 3), end="  ")
    for tw in (16, 8, 4):   # tile width in cells (256, 128, 64 px)
        n = tot = 0
        for cy in range(lv.shape[0]):
            for cx in range(0, lv.shape[1], tw):
                tot += 1; n += ok[cy, cx:cx + tw].all()
        print(f"tiles{tw*16}: {n/tot:.3f}", end="  ")
    print()
np.save(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "levels.npy"), lv)
