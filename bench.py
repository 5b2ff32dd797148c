#!/usr/bin/env python3
"""Benchmark of the hot path on the BASELINE.json metric configuration.

Metric: Mrays/s (+ ms/frame) of trace + filter on scene.obj, 1920x1080, 4 spp, maxBounces 5,
5 à-trous iterations (BASELINE.json configs[2]), on N GPUs of one node.
A "step" is one frame: 4 trace dispatches (sample k: time 1.25+0.0625k, blendFactor k/(k+1)),
5 filter passes, and for N > 1 the gather of the finished tiles to rank 0.  N > 1 is weak scaling by
default: the path shards by screen-space tiles, every rank owns one 1920x1080 tile (the metric's
configuration per GPU) of a frame that grows with N — 3840x1080 on 2, 3840x2160 (the resolution of
BASELINE.json configs[3]) on 4, 7680x2160 on 8 ranks —, traces it with the filter halo, filters it and
sends it to rank 0.  `--scaling strong` splits the fixed 1920x1080 frame instead.  All inputs are resident in
HBM before the timed region.  A ray = one closest-hit BVH traversal (primary, continuation or
shadow), counted by the device counters the parity tests check.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scene scene|room|...] [--size WxH]
  N > 1:  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
  python bench.py --scene lattice      SURVEY 8(d) config #5 (BASELINE.json configs[4]): 50.36 M triangles, 3840x2160,
                                       1 spp, maxBounces 4, no filter, traversal cap lifted — the HBM-bound regime
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
VALU_PEAK_TOPS_NOFMA = 78.6  # 157.3 TFLOP/s fp32 vector counts an FMA as 2; parity mode issues no FMA


def cpu_baseline(wl, spp, max_bounces, filter_iters, max_follows=1000):
    """The CPU restatement (oracle/, kind "port") timed on this host's cores on one frame of the
    same workload, bounded to ~10-30 s: rows are sub-sampled if a first probe says it would take longer."""
    from oracle import orclib
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    os.environ["OMP_NUM_THREADS"] = str(threads)
    sc = orclib.Scene(wl.buffers)
    W, H = wl.width, wl.height
    probe_rows = max(8, H // 32)
    y0 = (H - probe_rows) // 2
    t = time.time()
    orclib.trace(sc, W, H, wl.rays, wl.eye, spp=1, max_bounces=max_bounces, rect=(0, y0, W, y0 + probe_rows),
                 max_follows=max_follows)
    per_row = (time.time() - t) / probe_rows
    rows = H if per_row * H * spp < 25.0 else max(probe_rows, int(25.0 / (per_row * spp)))
    y0 = (H - rows) // 2
    t = time.time()
    o = orclib.trace(sc, W, H, wl.rays, wl.eye, spp=spp, max_bounces=max_bounces, rect=(0, y0, W, y0 + rows),
                     max_follows=max_follows)
    t_trace = time.time() - t
    rays = int(o["stats_sum"][..., 0].sum())
    t = time.time()
    if filter_iters:
        orclib.filter_passes(o["color"], o["normal"], o["pos"], filter_iters)
    t_filter = time.time() - t
    frame_s = t_trace * (H / rows) + t_filter
    return {"value": round(rays / t_trace / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "ms_per_frame": round(frame_s * 1e3, 1),
            "sample": f"oracle/rt_oracle.c (OpenMP, {threads} threads): {rows} of {H} rows x {W} px x {spp} spp "
                      f"traced in {t_trace:.2f} s, {filter_iters} filter passes on the full frame in {t_filter:.2f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="scene")
    ap.add_argument("--size", default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--bounces", type=int, default=None)
    ap.add_argument("--filter", type=int, default=None)
    ap.add_argument("--lattice", default="37x37x38", help="copies of suzanne.obj for --scene lattice")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = one --size tile per rank (the frame grows with N), strong = --size is the frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    lattice = a.scene == "lattice"
    # defaults: the metric configuration; for the lattice, config #5's
    a.size = a.size or ("3840x2160" if lattice else "1920x1080")
    a.spp = a.spp if a.spp is not None else (1 if lattice else 4)
    a.bounces = a.bounces if a.bounces is not None else (4 if lattice else 5)
    a.filter = a.filter if a.filter is not None else (0 if lattice else 5)
    max_follows = (1 << 30) if lattice else 1000
    W, H = (int(v) for v in a.size.lower().split("x"))

    import torch
    from raytracer_amd import hiplib, parallel, workloads

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit(f"--gpus {a.gpus} needs a torch.distributed launch with {a.gpus} ranks")
        a.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the hot path has no CPU fallback")
    # RT_BENCH_COLLECTIVES=gloo is a rehearsal of the N > 1 code path on a one-GPU box: every rank uses device 0 and
    # the strips travel through host memory.  The driver's runs use the default, RCCL ("nccl"), one GPU per rank.
    rehearsal = os.environ.get("RT_BENCH_COLLECTIVES", "nccl") == "gloo"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)

    tileW, tileH = W, H
    if world > 1 and a.scaling == "weak":
        # one W x H tile per rank: the frame is the tile grid's columns x rows of them
        halo = hiplib.lib().rt_filter_halo(a.filter) if a.filter > 0 else 0
        gx, gy = parallel.tile_grid(W, H, world, halo)
        W, H = W * gx, H * gy
    if lattice:
        nx, ny, nz = (int(v) for v in a.lattice.lower().split("x"))
        wl = workloads.LatticeWorkload(W, H, nx, ny, nz)
    else:
        wl = workloads.Workload(a.scene, W, H)
    ctx = hiplib.Context(local_rank)
    ctx.upload_scene(wl.buffers)
    ctx.resize(W, H)
    backend = parallel.HipBackend(ctx, wl, a.spp, a.bounces, a.filter, device, max_follows=max_follows)
    frame = parallel.TiledFrame(backend, W, H, rank, world,
                                hiplib.IMAGE_FILTERED if a.filter else hiplib.IMAGE_COLOR)

    def step():
        frame.render(0)
        frame.gather(dist)

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    ctx.reset_counters()
    ctx.profile(True)
    ctx.timings(reset=True)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0

    frame_sha = None
    if os.environ.get("RT_BENCH_CHECKSUM") and rank == 0:  # rehearsal aid: every N must assemble the same picture
        import hashlib
        img = ctx.readback(hiplib.IMAGE_FILTERED if a.filter else hiplib.IMAGE_COLOR)
        frame_sha = hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest()
    cnt = ctx.counters()
    tm = ctx.timings()
    # Tiles overlap by the filter halo, so the rays the ranks traced add up to more than the frame has.  `value`
    # counts the frame's own rays: every rank traces its tile once more WITHOUT the halo (outside the timed region;
    # a pixel's rays do not depend on the tile it is traced in) and the tiles partition the frame.
    frame_rays = float(cnt.rays) / max(a.steps, 1)
    if world > 1:
        ctx.profile(False)
        ctx.reset_counters()
        backend.trace(frame.tile, 0)
        ctx.sync()
        frame_rays = float(ctx.counters().rays)
    # Opt-in extra, outside the measured value: the same frames with RT_FLAG_FAST_SHADOWS (not the reference's
    # algorithm: shadow rays that miss their light are not cast, the others stop at the first nearer hit).
    fast_ms = None
    if world == 1:
        backend.flags = hiplib.FLAG_FAST_SHADOWS
        step()
        fence()
        t1 = time.perf_counter()
        for _ in range(max(1, a.steps // 2)):
            step()
        fence()
        fast_ms = (time.perf_counter() - t1) * 1e3 / max(1, a.steps // 2)
        backend.flags = 0
    K = hiplib
    local = torch.tensor([elapsed, float(cnt.rays), float(cnt.node_visits), float(cnt.tri_tests),
                          float(cnt.shaded_hits), float(cnt.paths)] + [tm.ms[i] for i in range(8)] + [frame_rays],
                         dtype=torch.float64, device="cpu" if rehearsal else device)
    if dist is not None:
        tmax = local.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = local.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        sums = tsum.tolist()
        maxs = tmax.tolist()
    else:
        sums = maxs = local.tolist()
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    rays, visits, tests, hits, paths = sums[1:6]
    frame_rays = sums[14]
    ms_per_step = elapsed * 1e3 / a.steps
    mrays = frame_rays * a.steps / elapsed / 1e6
    # Roofline of the dominant kernel, per launch, from HIP events recorded on the kernel's own stream
    # (rt_profile).  The dominant kernel is whichever of the pipeline's kernels took most device time on
    # this rank; its algorithmic bytes follow SURVEY.md 8(d): 48 B per node visit + 48 B per triangle
    # test it performed (+ 160 B per shaded hit and the per-pixel-sample image bytes for kernels that do those).
    kernel_ms = {K.TIME_NAMES[i]: tm.ms[i] for i in range(3, 8) if tm.launches[i]}
    dominant = max(kernel_ms, key=kernel_ms.get)
    didx = K.TIME_NAMES.index(dominant)
    launches = int(tm.launches[didx])
    avg_launch_ms = tm.ms[didx] / max(launches, 1)
    primary_rays = float(cnt.paths)
    sec_visits, sec_tests = cnt.node_visits - cnt.primary_node_visits, cnt.tri_tests - cnt.primary_tri_tests
    if dominant == "wf_extend" and lattice:
        # scenes beyond the L2s: the primary rays that enter the scene box are walked by the extend kernel too
        # (stage 0), so all visits and tests are its work (a ray that misses the box costs the primary kernel one visit)
        alg_bytes = 48.0 * (cnt.node_visits + cnt.tri_tests) + 48.0 * cnt.rays
        ops = 25.0 * cnt.node_visits + 51.0 * cnt.tri_tests
        kname = "rt_wf_extend_kernel"
    elif dominant == "wf_extend":  # every non-primary traversal; reads a 32 B ray, writes a 16 B hit per ray
        alg_bytes = 48.0 * (sec_visits + sec_tests) + 48.0 * (cnt.rays - primary_rays)
        ops = 25.0 * sec_visits + 51.0 * sec_tests
        kname = "rt_wf_extend_kernel"
    elif dominant == "per_pixel":
        alg_bytes = (48.0 * (cnt.node_visits + cnt.tri_tests) + 160.0 * cnt.shaded_hits + 28.0 * cnt.paths +
                     16.0 * cnt.paths * (a.spp - 1) / max(a.spp, 1))
        ops = 25.0 * cnt.node_visits + 51.0 * cnt.tri_tests
        kname = "rt_trace3_kernel"
    elif dominant == "wf_primary":  # primary traversals + G-buffer / radiance writes (28 B per pixel-sample)
        alg_bytes = 48.0 * (cnt.primary_node_visits + cnt.primary_tri_tests) + 28.0 * cnt.paths
        ops = 25.0 * cnt.primary_node_visits + 51.0 * cnt.primary_tri_tests
        kname = "rt_wf_primary_kernel"
    else:                           # logic / blend: path-state streaming, ~300 B per queued path and stage
        alg_bytes = 160.0 * cnt.shaded_hits + 300.0 * cnt.shaded_hits
        ops = 0.0
        kname = "rt_" + dominant + "_kernel"
    alg_bytes_per_launch = alg_bytes / max(launches, 1)
    achieved = alg_bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    tops = ops / max(launches, 1) / (avg_launch_ms * 1e-3) / 1e12 if avg_launch_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                traffic = json.load(f).get(f"{wl.name if lattice else a.scene}_{W}x{H}_{kname}_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "Mrays/s (trace+filter), scene.obj 1080p 4spp" if (a.scene, tileW, tileH, a.spp) == ("scene", 1920, 1080, 4)
                  else f"Mrays/s (trace+filter), {wl.name if lattice else a.scene} {W}x{H} {a.spp}spp",
        "value": round(mrays, 3), "unit": "Mrays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 4), "ms_per_frame": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        **({"rehearsal": "gloo collectives, all ranks on GPU 0: not a measurement"} if rehearsal else {}),
        "config": {"workload": (f"{wl.info.num_triangles} triangles: suzanne.obj on a jittered {a.lattice} lattice (PCG32 seed "
                                f"0x5EED), {W}x{H} {a.spp}spp maxBounces {a.bounces}, {a.filter} filter iterations, "
                                f"MAX_FOLLOWS lifted (BASELINE.json configs[4], SURVEY 8(d) config #5)") if lattice else
                               (f"{a.scene}.obj {W}x{H} {a.spp}spp maxBounces {a.bounces} + {a.filter}-iteration "
                                f"a-trous filter, SURVEY 8(d) auto camera + harness quad light"
                                + (f" (BASELINE.json configs[2] = {tileW}x{tileH} per GPU)" if (W, H) != (tileW, tileH)
                                   else " (BASELINE.json configs[2])")),
                   "triangles": wl.info.num_triangles, "nodes": wl.info.num_nodes, "emitters": wl.num_emitters,
                   "schedule": "wavefront pipeline, samples of a frame traced concurrently (rt_trace_batch)",
                   "frame": f"{W}x{H}",
                   "parallelism": f"{world} ranks as a {frame.grid[0]} x {frame.grid[1]} grid of screen-space tiles"
                                  + (f" of {tileW}x{tileH} (the frame grows with N)" if world > 1 and a.scaling == "weak" else "")
                                  + f", halo {backend.halo} px, one gather to rank 0"},
        "rays_per_frame": frame_rays, "rays_traced_per_frame_incl_halo": rays / a.steps,
        "node_visits_per_frame": visits / a.steps,
        "tri_tests_per_frame": tests / a.steps,
        "kernel_ms_per_frame_max_rank": {K.TIME_NAMES[i]: round(maxs[6 + i] / a.steps, 4) for i in range(8)
                                         if maxs[6 + i] > 0},
        "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": traffic, "avg_launch_ms": round(avg_launch_ms, 4), "launches": launches,
                     "algorithmic_bytes_per_launch": round(alg_bytes_per_launch),
                     "note": ("scene + BVH are beyond every cache: the algorithmic bytes are HBM bytes (traffic = measured "
                              "HBM bytes per launch)") if lattice else
                             ("scene + BVH (<1 MB) stay resident in every XCD's L2, so these algorithmic bytes are "
                              "served by L1/L2 and not by HBM (traffic = measured HBM bytes); the kernel is limited by "
                              "vector-L1 gather throughput, fp32 VALU issue and divergence (see valu)"),
                     "valu": {"achieved_Tops": round(tops, 4), "peak_Tops_no_fma": VALU_PEAK_TOPS_NOFMA,
                              "frac": round(tops / VALU_PEAK_TOPS_NOFMA, 4),
                              "ops_model": "25 per node visit + 51 per triangle test (SURVEY 8(d))"}},
    }
    if frame_sha is not None:
        out["frame_sha256"] = frame_sha
    if fast_ms is not None:
        out["opt_in"] = {"fast_shadows_ms_per_frame": round(fast_ms, 4),
                         "note": "RT_FLAG_FAST_SHADOWS: light-first any-hit shadow rays (SURVEY 8(f) rank 4); same image up "
                                 "to exact-distance ties, fewer rays traced; not part of `value`"}
    if world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(wl, a.spp, a.bounces, a.filter, max_follows)
        except Exception as e:  # the checker library is test infrastructure; never fatal for the measurement
            out["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "port",
                                   "sample": f"unavailable: {e}"}
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
