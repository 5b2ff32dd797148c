"""Development aid: radiance agreement of the HIP trace path with the llvmpipe fixtures and the CPU oracle
(outlier fraction beyond tests/util.py's per-pixel tolerance, RMSE / mean, max relative difference).
usage: python scripts/radiance_margin.py [path of an alternative librt_hip.so]"""
import glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import hiplib, workloads
if len(sys.argv) > 1:
    hiplib.LIB_PATH = os.path.abspath(sys.argv[1])
from tests.util import GOLDEN, load_npz, radiance_report
from oracle import orclib
ctx = hiplib.Context(0)

def render(wl, spp, mb):
    ctx.upload_scene(wl.buffers); ctx.resize(wl.width, wl.height)
    ctx.trace_samples(wl.rays, wl.eye, spp=spp, max_bounces=mb, flags=hiplib.FLAG_PIXEL_STATS, num_emitters=wl.num_emitters)
    return ctx.readback(hiplib.IMAGE_COLOR), ctx.readback(hiplib.IMAGE_STATS)

for p in sorted(glob.glob(os.path.join(GOLDEN, "trace_*.npz"))):
    g = load_npz(os.path.basename(p)); m = g["meta"]
    wl = workloads.Workload(m["scene"], m["width"], m["height"])
    col, st = render(wl, m["spp"], m["max_bounces"])
    frac, rmse, mean, mx = radiance_report(g["color"], col)
    print(f"{os.path.basename(p):44s} outliers {frac:.2e} rmse/mean {rmse / max(mean, 1e-3):.2e} max rel {mx:.2e} rays differ {(st[..., 0] != g['stats'][..., 0]).mean():.2e}")
for scene, w, h, spp, mb in [("room", 400, 225, 4, 5), ("scene", 1920, 1080, 4, 5), ("room", 800, 450, 2, 5)]:
    wl = workloads.Workload(scene, w, h)
    col, st = render(wl, spp, mb)
    ref = orclib.trace(orclib.Scene(wl.buffers), w, h, wl.rays, wl.eye, spp=spp, max_bounces=mb)
    frac, rmse, mean, mx = radiance_report(ref["color"], col)
    print(f"oracle {scene} {w}x{h} {spp}spp mb{mb}:".ljust(44) + f" outliers {frac:.2e} rmse/mean {rmse / max(mean, 1e-3):.2e} max rel {mx:.2e} rays differ {(st[..., 0] != ref['stats'][..., 0]).mean():.2e}")
