"""Development aid: extend-kernel time against the number of LDS-cached nodes (RT_WF_CACHE_NODES)."""
import os, subprocess, sys
sys.path.insert(0, '.')
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from scripts.gpu_perf import run
    from raytracer_amd import hiplib, workloads
    run('scene', 1920, 1080)
    run('room', 1920, 1080)
    import time
    wl = workloads.LatticeWorkload(1920, 1080, 12, 12, 12)
    ctx = hiplib.Context(0); ctx.upload_scene(wl.buffers); ctx.resize(1920, 1080)
    for rep in range(3):
        ctx.reset_counters(); ctx.profile(True); ctx.timings(reset=True)
        ctx.trace_samples(wl.rays, wl.eye, spp=1, max_bounces=4, num_emitters=wl.num_emitters, max_follows=1 << 30); ctx.sync()
    print("      lattice12 1080p 1spp:", {k: round(v[0], 3) for k, v in ctx.timings().as_dict().items()}, flush=True)
else:
    for n in sys.argv[1:] or ["0", "256", "900"]:
        print("=== RT_WF_CACHE_NODES", n, flush=True)
        env = dict(os.environ, RT_WF_CACHE_NODES=n)
        subprocess.run([sys.executable, "-u", "scripts/gpu_cache_sweep.py", "child"], env=env, check=True, timeout=200)
