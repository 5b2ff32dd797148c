"""Ad-hoc GPU check: HIP trace3 vs CPU oracle on the named scenes (development aid)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from raytracer_amd import hiplib, workloads
from oracle import orclib

def compare(name, W, H, spp, mb, filt=0):
    wl = workloads.Workload(name, W, H)
    ctx = hiplib.Context(0)
    ctx.upload_scene(wl.buffers)
    ctx.resize(W, H)
    ctx.profile(True)
    flags = hiplib.FLAG_AUX | hiplib.FLAG_PIXEL_STATS
    t = time.time()
    ctx.trace_samples(wl.rays, wl.eye, spp=spp, max_bounces=mb, flags=flags)
    ctx.sync(); tg = time.time() - t
    g = {k: ctx.readback(v) for k, v in dict(color=hiplib.IMAGE_COLOR, normal=hiplib.IMAGE_NORMAL, pos=hiplib.IMAGE_POSITION, tridx=hiplib.IMAGE_TRIDX, visits=hiplib.IMAGE_VISITS, stats=hiplib.IMAGE_STATS).items()}
    cnt = ctx.counters(); tm = ctx.timings()
    sc = orclib.Scene(wl.buffers)
    t = time.time()
    o = orclib.trace(sc, W, H, wl.rays, wl.eye, spp=spp, max_bounces=mb)
    to = time.time() - t
    print(f"== {name} {W}x{H} spp{spp} mb{mb}: gpu wall {tg*1e3:.1f} ms, kernel {tm.trace_ms:.3f} ms/{tm.trace_launches}; oracle {to:.2f}s; rays {cnt.rays} visits {cnt.node_visits} tests {cnt.tri_tests} paths {cnt.paths}")
    print("   tridx mismatch", int((g['tridx'] != o['tridx']).sum()), "visits mismatch", int((g['visits'] != o['visits']).sum()),
          "normal", int((g['normal'] != o['normal']).sum()), "pos", int((g['pos'].view(np.uint16) != o['pos'].view(np.uint16)).sum()),
          "hit px", int((o['tridx'] >= 0).sum()))
    a, b = o['color'][..., :3].astype(np.float64), g['color'][..., :3].astype(np.float64)
    nan = np.isnan(a).any(2) | np.isnan(b).any(2)
    rel = np.abs(a - b) / np.maximum(np.abs(a), 1e-3); rel[nan] = 0
    print("   colour: nan orc", int(np.isnan(a).any(2).sum()), "gpu", int(np.isnan(b).any(2).sum()), "max rel", rel.max(), "px rel>1e-3", int((rel.max(2) > 1e-3).sum()),
          "rmse", np.sqrt(np.mean((a - b)[~nan] ** 2)), "mean", a[~nan].mean())
    so, sg = o['stats'][..., :3].reshape(-1, 3).sum(0), g['stats'][..., :3].reshape(-1, 3).sum(0)
    print("   stats sums orc", so, "gpu", sg, "px differing", int((o['stats'][..., :3] != g['stats'][..., :3]).any(2).sum()))
    if filt:
        ctx.filter(filt); ctx.sync()
        gf = ctx.readback(hiplib.IMAGE_FILTERED)
        of = orclib.filter_passes(g['color'], g['normal'], g['pos'], filt)[-1]
        d = np.abs(gf.view(np.uint16).astype(np.int32) - of.view(np.uint16).astype(np.int32))
        print("   filter: exact", (d == 0).mean(), "<=1ulp", (d <= 1).mean(), "max ulp", d.max(), "kernel ms", ctx.timings().filter_ms)
    ctx.close()

compare('room', 160, 120, 1, 0)
compare('room', 160, 120, 2, 5, filt=5)
compare('suzanne', 320, 180, 2, 5, filt=5)
compare('scene', 320, 180, 2, 5, filt=5)
compare('teapot', 320, 240, 1, 5)
