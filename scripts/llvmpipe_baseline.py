"""Times the REFERENCE shaders on Mesa llvmpipe on the metric configuration (build container only:
needs /root/reference).  Writes profiles/llvmpipe_baseline.json.  SURVEY.md 8(d) "CPU baseline"."""
import json, os, sys, time
sys.path.insert(0, '.')
import numpy as np
from oracle import refrun
from raytracer_amd import workloads

def run(name, W, H, spp, mb, iters, repeat=3):
    wl = workloads.Workload(name, W, H)
    r = refrun.run(wl.buffers, W, H, wl.rays, wl.eye, spp=spp, max_bounces=mb, num_emitters=wl.num_emitters,
                   instrument=False, filter_iters=iters, repeat=repeat)
    tr = np.array(r["trace_ms"]).reshape(repeat, spp)
    frames = tr[1:].sum(1) if repeat > 1 else tr.sum(1)      # first repetition includes the LLVM JIT
    fl = r["filter_ms"][1:] if repeat > 1 else r["filter_ms"]
    ri = refrun.run(wl.buffers, W, H, wl.rays, wl.eye, spp=1, max_bounces=mb, num_emitters=wl.num_emitters,
                    instrument=True)
    rays = int(ri["stats"][..., 0].sum()) * spp
    out = dict(scene=name, width=W, height=H, spp=spp, max_bounces=mb, filter_iters=iters,
               trace_ms_per_frame=float(np.median(frames)), filter_ms_per_frame=float(np.median(fl)) if fl else 0.0,
               rays_per_frame=rays, mrays_per_s=rays / (float(np.median(frames)) * 1e3),
               cores=os.cpu_count(), gl=[l for l in r["log"].splitlines() if l.startswith("GL_")])
    print(json.dumps(out), flush=True)
    return out

if __name__ == "__main__":
    res = [run("scene", 1920, 1080, 4, 5, 5), run("room", 1920, 1080, 4, 5, 5, repeat=2), run("teapot", 640, 480, 1, 5, 0)]
    os.makedirs("profiles", exist_ok=True)
    json.dump(res, open("profiles/llvmpipe_baseline.json", "w"), indent=1)
