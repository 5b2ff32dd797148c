import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from raytracer_amd import hiplib, workloads
for scene in ("scene", "room"):
    W, H = 1920, 1080
    wl = workloads.Workload(scene, W, H)
    ctx = hiplib.Context(0); ctx.upload_scene(wl.buffers); ctx.resize(W, H)
    def frame(): ctx.trace_samples(wl.rays, wl.eye, spp=4, max_bounces=5, num_emitters=wl.num_emitters, flags=hiplib.FLAG_PER_PIXEL_KERNEL)
    frame(); ctx.sync()
    t = time.time()
    for _ in range(3): frame()
    ctx.sync()
    print(scene, "per-pixel kernel ms per 4-spp frame", round((time.time() - t) / 3 * 1e3, 2))
    ctx.close()
