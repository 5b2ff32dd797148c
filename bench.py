#!/usr/bin/env python3
"""Benchmark of the hot path on the BASELINE.json metric configuration.

Metric: Mrays/s (+ ms/frame) of trace + filter on scene.obj, 1920x1080, 4 spp, maxBounces 5,
5 à-trous iterations (BASELINE.json configs[2]), on N GPUs of one node.
A "step" is one frame: 4 trace dispatches (sample k: time 1.25+0.0625k, blendFactor k/(k+1)),
5 filter passes, and for N > 1 the gather of the finished tiles to rank 0.  All inputs are resident in
HBM before the timed region.  A ray = one closest-hit BVH traversal (primary, continuation or
shadow) that was actually executed, counted by the device counters the parity tests check.

N > 1 is STRONG scaling of that fixed frame by default (BASELINE.json quotes the metric "at 1/2/4/8 GPUs"): the
frame is split into a columns x rows grid of screen-space tiles, every rank traces its tile plus the filter halo,
filters it locally and takes part in one gather (rt_gather_tiles: RCCL send/recv on a second stream, overlapped
with the next frame's trace).  `--scaling weak` gives every rank one --size tile of a frame that grows with N
instead; the metric string always names the frame that was rendered.

With N == 1 the same JSON line also carries, under "cases", the two configurations in which the kernels are
throughput- rather than latency-bound, each with its own ms_per_frame, roofline and cpu_baseline:
  cases.room     room.obj (the reference's own scene family and camera: every pixel sees geometry), 1080p 4 spp + filter
  cases.lattice  SURVEY 8(d) config #5 (BASELINE.json configs[4]): 50.36 M triangles, 3840x2160, 1 spp, maxBounces 4,
                 traversal cap lifted — the HBM-bound regime
With N > 1 it carries cases.config4 (BASELINE.json configs[3]: scene.obj 3840x2160 16 spp, strong-scaled over the N ranks).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scene scene|room|lattice|...] [--size WxH] [--cases room,lattice|none]
  N > 1:  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Eight hardware queues for this process instead of HIP's default four (read once, when the HIP runtime starts): only the
# `steady_state` leg below needs them — its eight lanes would otherwise share four queues and run in pairs one after the
# other.  The driver-timed measurement runs four lanes and is the same with either setting (measured: 0.773 / 0.779 ms).
if int(os.environ.get("GPU_MAX_HW_QUEUES") or 4) < 8:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"

THROUGHPUT_NOTE = ("throughput: wall time of the K timed frames, queued back to back with `frames_in_flight` frames "
                   "overlapping on the GPU, divided by K; `latency_ms_one_frame` is one frame on an otherwise idle GPU")
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
VALU_PEAK_TOPS_NOFMA = 78.6  # 157.3 TFLOP/s fp32 vector counts an FMA as 2; parity mode issues no FMA
L2_RESIDENT_BYTES = 3 * 1024 * 1024  # the launcher's own threshold (rt_launcher.hip): hot scene bytes vs one XCD's 4 MiB L2


def cpu_baseline(wl, spp, max_bounces, filter_iters, max_follows=1000, budget_s=20.0):
    """The CPU restatement (oracle/, kind "port") timed on this host's cores on one frame of the
    same workload, bounded to about `budget_s`: rows are sub-sampled if a first probe says it would take longer."""
    from oracle import orclib
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    os.environ["OMP_NUM_THREADS"] = str(threads)
    sc = orclib.Scene(wl.buffers)
    W, H = wl.width, wl.height
    probe_rows = max(8, H // 64)
    y0 = (H - probe_rows) // 2
    t = time.time()
    orclib.trace(sc, W, H, wl.rays, wl.eye, spp=1, max_bounces=max_bounces, rect=(0, y0, W, y0 + probe_rows),
                 max_follows=max_follows)
    per_row = (time.time() - t) / probe_rows
    rows = H if per_row * H * spp < budget_s else max(probe_rows, int(budget_s / (per_row * spp)))
    y0 = (H - rows) // 2
    # a frame that is mostly sky takes a fraction of a second (the probe rows are the busiest ones): repeat it until a
    # fifth of the budget has passed on the wall clock, i.e. 16 threads x 3-5 s of CPU work
    repeats = 0
    t = time.time()
    while True:
        o = orclib.trace(sc, W, H, wl.rays, wl.eye, spp=spp, max_bounces=max_bounces, rect=(0, y0, W, y0 + rows),
                         max_follows=max_follows)
        repeats += 1
        if rows < H or repeats >= 256 or time.time() - t >= 0.2 * budget_s:
            break
    t_trace = (time.time() - t) / repeats
    rays = int(o["stats_sum"][..., 0].sum())
    t = time.time()
    if filter_iters:
        orclib.filter_passes(o["color"], o["normal"], o["pos"], filter_iters)
    t_filter = time.time() - t
    frame_s = t_trace * (H / rows) + t_filter
    return {"value": round(rays / t_trace / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "ms_per_frame": round(frame_s * 1e3, 1),
            "sample": f"oracle/rt_oracle.c (OpenMP, {threads} threads): the middle {rows} of {H} rows x {W} px x {spp} spp "
                      f"traced {repeats}x in {t_trace:.2f} s each, {filter_iters} filter passes on the full frame in {t_filter:.2f} s"}


def alone_roofline(K, ctx, step, steps, wl, W, H, spp, filter_iters, scene_key, root_misses):
    """The timed region runs with several frames in flight (four by default), so its kernel spans are the durations of launches that
    share the GPU with the other frame's.  This measures the same kernel once more with one frame at a time
    (outside the timed region): what a launch that has the GPU to itself achieves."""
    ctx.sync()
    before = ctx.frames_in_flight()
    ctx.set_frames_in_flight(1)
    try:
        step()
        ctx.sync()
        ctx.reset_counters()
        ctx.profile(True)
        ctx.timings(reset=True)
        for _ in range(steps):
            step()
        ctx.sync()
        cnt, tm = ctx.counters(), ctx.timings()
        ctx.profile(False)
        r = roofline(K, tm, cnt, wl, W, H, spp, filter_iters, steps, scene_key, root_misses)
    finally:
        ctx.set_frames_in_flight(before if os.environ.get("RT_PIPELINE") else 0)  # 0: back to the launcher's choice
    if not r:
        return None
    return {"frames_in_flight_of_timed_region": before, **{k: r[k] for k in ("kernel", "achieved", "unit", "frac", "avg_launch_ms", "launches")}}


def one_frame_latency(ctx, step, frames=10):
    """Wall-clock time of ONE frame (trace + filter) from its first launch to its last kernel's end with nothing else on
    the GPU: frames in flight set to 1, a sync before and after every frame, no profiling events.  Median of `frames`."""
    ctx.sync()
    before = ctx.frames_in_flight()
    ctx.set_frames_in_flight(1)
    try:
        step()
        ctx.sync()
        times = []
        for _ in range(frames):
            ctx.sync()
            t0 = time.perf_counter()
            step()
            ctx.sync()
            times.append((time.perf_counter() - t0) * 1e3)
        # throughput with one frame at a time: the same frames queued back to back, one sync at the end
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(frames):
            step()
        ctx.sync()
        back_to_back = (time.perf_counter() - t0) * 1e3 / frames
    finally:
        ctx.set_frames_in_flight(before if os.environ.get("RT_PIPELINE") else 0)
    return round(float(np.median(times)), 4), round(back_to_back, 4)


def steady_state(ctx, step, lanes=8, frames=120):
    """A long run of frames with `lanes` frames in flight (rt_set_frames_in_flight; needs as many hardware queues): what a
    renderer that never synchronises gets.  The driver-timed region is a burst of K frames between two syncs, where the
    cost of filling and draining the lanes is a visible part and more than four lanes do not pay."""
    ctx.sync()
    before = ctx.frames_in_flight()
    ctx.set_frames_in_flight(lanes)
    try:
        for _ in range(2 * lanes):
            step()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(frames):
            step()
        ctx.sync()
        ms = (time.perf_counter() - t0) * 1e3 / frames
        got = ctx.frames_in_flight()
    finally:
        ctx.set_frames_in_flight(before if os.environ.get("RT_PIPELINE") else 0)
    return {"frames_in_flight": got, "frames": frames, "ms_per_frame": round(ms, 4),
            "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES") or 4),
            "note": "not `value`: the same frames in one run of 120 with eight lanes (the process is given eight hardware queues); "
                    "a burst of 20 between two syncs, as the driver times it, gains nothing from more than four"}


def with_alone(rl, alone):
    if rl is not None:
        rl["frames_in_flight"] = alone.pop("frames_in_flight_of_timed_region") if alone else None
        rl["overlap_note"] = ("several frames in flight: a launch shares the GPU with the other frames' kernels, so its span "
                              "is longer than its cost; one_frame_at_a_time = the same kernel with the GPU to itself")
        rl["one_frame_at_a_time"] = alone
    return rl


class _GlooGather:
    """torch.distributed.gather over a gloo group (host memory): the exchange bench.py falls back to when the
    launcher's RCCL communicator cannot be set up."""

    def __init__(self, dist):
        self.dist = dist
        self.group = dist.new_group(backend="gloo")

    def gather(self, tensor, gather_list=None, dst=0):
        self.dist.gather(tensor, gather_list, dst=dst, group=self.group)


def roofline(K, tm, cnt, wl, W, H, spp, filter_iters, steps, scene_key, root_misses=0):
    """Roofline of the kernel that took most device time, per launch, from HIP events recorded on the kernel's own
    stream (rt_profile).  Algorithmic work follows SURVEY.md 8(d): per traversal 25 ops + 48 B per node visit,
    51 ops + 48 B per triangle test; filter 36 B (pass 0) / 28 B per pixel and pass.  Which roof applies is decided
    by where the scene lives: hot scene bytes (32 B per node + 48 B per triangle) within one XCD's L2 -> the trace
    kernels are bound by fp32 VALU issue (no FMA in parity mode: 78.6 Tops/s), their algorithmic bytes are served by
    LDS / L1 / L2 and `traffic` (measured HBM bytes) is orders of magnitude below them; beyond the L2s -> HBM."""
    names = K.TIME_NAMES
    kernel_ms = {names[i]: tm.ms[i] for i in (1, 3, 4, 5, 6, 7, 8) if tm.launches[i]}
    if not kernel_ms:
        return None
    dominant = max(kernel_ms, key=kernel_ms.get)
    didx = names.index(dominant)
    launches = int(tm.launches[didx])
    avg_ms = tm.ms[didx] / max(launches, 1)
    hot = 32 * wl.info.num_nodes + 48 * wl.info.num_triangles
    resident = hot <= L2_RESIDENT_BYTES
    # Traversals run in the extend kernels (stage 0 = the primary rays that enter the scene's bounding box, traced once
    # per pixel); the classify kernel only does the root-box test: a primary ray that misses that box is one ray of one
    # node visit, counted there (root_misses per frame).
    classified = float(root_misses) * steps
    ext_rays, ext_visits, ext_tests = cnt.rays - classified, cnt.node_visits - classified, float(cnt.tri_tests)
    ops = 0.0
    # (the stages that hold few rays run in rt_wf_extend_rows_kernel: timed and counted on their own)
    if dominant in ("wf_extend", "wf_extend_rows"):    # reads a 32 B ray, writes a 16 B hit per ray
        if dominant == "wf_extend_rows":
            ext_rays, ext_visits, ext_tests = float(cnt.rows_rays), float(cnt.rows_node_visits), float(cnt.rows_tri_tests)
        else:
            ext_rays, ext_visits, ext_tests = (ext_rays - cnt.rows_rays, ext_visits - cnt.rows_node_visits,
                                               ext_tests - cnt.rows_tri_tests)
        alg_bytes = 48.0 * (ext_visits + ext_tests) + 48.0 * ext_rays
        ops = 25.0 * ext_visits + 51.0 * ext_tests
        kname = "rt_" + dominant + "_kernel"
    elif dominant == "per_pixel":
        alg_bytes = (48.0 * (cnt.node_visits + cnt.tri_tests) + 160.0 * cnt.shaded_hits + 28.0 * cnt.paths +
                     16.0 * cnt.paths * (spp - 1) / max(spp, 1))
        ops = 25.0 * cnt.node_visits + 51.0 * cnt.tri_tests
        kname = "rt_trace3_kernel"
    elif dominant == "wf_primary":  # root-box test per pixel + G-buffer / radiance writes for the pixels that miss it
        alg_bytes = 48.0 * classified + 28.0 * classified + 32.0 * (float(W * H) * steps - classified)
        ops = 25.0 * float(W * H) * steps
        kname = "rt_wf_classify_kernel"
    elif dominant == "filter":     # 36 B per pixel in pass 0, 28 B in later passes
        alg_bytes = float(W * H) * (36.0 + 28.0 * (filter_iters - 1)) * steps
        kname = "rt_atrous_kernel"
    else:                           # logic / blend: path-state streaming, ~300 B per queued path and stage
        alg_bytes = 160.0 * cnt.shaded_hits + 300.0 * cnt.shaded_hits
        kname = "rt_" + dominant + "_kernel"
    bytes_per_launch = alg_bytes / max(launches, 1)
    gbs = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    tops = ops / max(launches, 1) / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    traffic = memside = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                tj = json.load(f)
            traffic = tj.get(f"{scene_key}_{W}x{H}_{kname}_bytes_per_launch")
            memside = tj.get(f"{scene_key}_{W}x{H}_{kname}_memside")
        except Exception:
            traffic = None
    valu_bound = resident and ops > 0
    r = {"bound": "valu" if valu_bound else "hbm", "kernel": kname,
         "achieved": round(tops if valu_bound else gbs, 4 if valu_bound else 2),
         "peak": VALU_PEAK_TOPS_NOFMA if valu_bound else HBM_PEAK_GBS,
         "unit": "TFLOP/s" if valu_bound else "GB/s",
         "frac": round(tops / VALU_PEAK_TOPS_NOFMA if valu_bound else gbs / HBM_PEAK_GBS, 4),
         "traffic": traffic,
         "traffic_source": ("profiles/traffic.json: rocprofv3 --pmc passes of this command on the committed build with one frame at a "
                            "time (scripts/profile.sh, scripts/pmc_memside.sh) - PMC counters cannot be collected inside a timed run; "
                            "bytes the L2 requested from the fabric (Infinity-Cache hits included), per launch"),
         "avg_launch_ms": round(avg_ms, 4), "launches": launches,
         "algorithmic_bytes_per_launch": round(bytes_per_launch),
         "algorithmic_ops_per_launch": round(ops / max(launches, 1)),
         "ops_model": "25 fp32 ops per node visit + 51 per triangle test, FMA not used (SURVEY 8(d))",
         "hot_scene_bytes": hot,
         "note": ("hot scene bytes fit every XCD's 4 MiB L2: algorithmic bytes are served by LDS / L1 / L2 (traffic = "
                  "measured HBM bytes per launch, far below them), the roof is fp32 VALU issue without FMA") if valu_bound
                 else ("scene + BVH are beyond every cache: the algorithmic bytes are HBM bytes (traffic = measured HBM "
                       "bytes per launch)" if not resident else "streaming kernel: algorithmic bytes against the HBM roof")}
    if not valu_bound and ops > 0:
        r["valu"] = {"achieved_Tops": round(tops, 4), "frac": round(tops / VALU_PEAK_TOPS_NOFMA, 4)}
    if not valu_bound and not resident:
        # `achieved` is ALGORITHMIC bytes per second.  Part of them is served by the L2s and, of the L2 misses, part by the
        # 256 MiB Infinity Cache: what reaches the HBM is less, and no gfx950 counter offered by rocprofv3 separates the two.
        r["achieved_is"] = ("algorithmic bytes / launch time (L2 hits and Infinity-Cache hits included): an upper bound of the "
                            "HBM rate, not the HBM rate")
        if memside:
            r["memory_side"] = memside
            if avg_ms > 0 and memside.get("l2_miss_read_bytes_per_launch"):
                r["memory_side"]["l2_miss_GBs_at_this_launch_time"] = round(
                    (memside["l2_miss_read_bytes_per_launch"] + memside.get("write_bytes_per_launch", 0)) / (avg_ms * 1e-3) / 1e9, 1)
    if valu_bound:
        r["algorithmic_GBs"] = round(gbs, 2)
    return r


def frame_statistics(K, ctx, backend, wl, W, H, spp):
    """How much of the frame sees geometry (one untimed 1-spp, 0-bounce frame with the id / visit images)."""
    ctx.trace_samples(wl.rays, wl.eye, spp=1, max_bounces=0, flags=K.FLAG_AUX, num_emitters=wl.num_emitters,
                      max_follows=backend.max_follows)
    tri, vis = ctx.readback(K.IMAGE_TRIDX), ctx.readback(K.IMAGE_VISITS)
    hit = float((tri >= 0).mean())
    stop_at_root = int(((tri < 0) & (vis <= 1)).sum())
    return hit, stop_at_root


def run_case(K, parallel, ctx, device, wl, scene_key, W, H, spp, bounces, filter_iters, max_follows, steps, warmup,
             cpu_budget_s):
    """One single-GPU case: upload, settle the launcher's schedule, W warm-up frames, K timed frames."""
    import torch
    ctx.upload_scene(wl.buffers)
    ctx.resize(W, H)
    backend = parallel.HipBackend(ctx, wl, spp, bounces, filter_iters, device, max_follows=max_follows)
    hit, stop_at_root = frame_statistics(K, ctx, backend, wl, W, H, spp)
    ctx.resize(W, H)
    rect = (0, 0, W, H)

    def step():
        backend.trace(rect, 0)
        backend.filter(rect)

    settle = 0
    while ctx.walk_stop() == 0 and settle < 6:  # the launcher times its two walk schedules on the first frames (set-up)
        step()
        settle += 1
    for _ in range(warmup):
        step()
    ctx.sync()
    ctx.reset_counters()
    ctx.profile(True)
    ctx.timings(reset=True)
    ctx.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    cnt, tm = ctx.counters(), ctx.timings()
    ctx.profile(False)
    rays_per_frame = float(cnt.rays) / steps
    in_flight = ctx.frames_in_flight()
    rl = with_alone(roofline(K, tm, cnt, wl, W, H, spp, filter_iters, steps, scene_key, stop_at_root),
                    alone_roofline(K, ctx, step, steps, wl, W, H, spp, filter_iters, scene_key, stop_at_root))
    latency, alone_b2b = one_frame_latency(ctx, step, max(3, min(10, steps)))
    out = {"ms_per_frame": round(elapsed * 1e3 / steps, 4), "ms_per_frame_is": THROUGHPUT_NOTE, "frames_in_flight": in_flight,
           "latency_ms_one_frame": latency, "ms_per_frame_one_frame_at_a_time": alone_b2b,
           "steps": steps, "warmup": warmup,
           "value": round(float(cnt.rays) / elapsed / 1e6, 3), "unit": "Mrays/s",
           "rays_per_frame": rays_per_frame, "node_visits_per_frame": cnt.node_visits / steps,
           "tri_tests_per_frame": cnt.tri_tests / steps,
           "hit_pixel_fraction": round(hit, 5),
           "rays_past_root_per_frame": rays_per_frame - float(stop_at_root),
           "rays_per_frame_if_primary_per_sample": rays_per_frame + float((spp - 1) * W * H),
           "walk_stop": ctx.walk_stop(), "schedule_settle_frames": settle,
           "kernel_ms_per_frame": {K.TIME_NAMES[i]: round(tm.ms[i] / steps, 4) for i in range(len(K.TIME_NAMES)) if tm.launches[i]},
           "roofline": rl}
    if cpu_budget_s > 0:
        try:
            out["cpu_baseline"] = cpu_baseline(wl, spp, bounces, filter_iters, max_follows, cpu_budget_s)
        except Exception as e:  # the checker library is test infrastructure; never fatal for the measurement
            out["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    return out, backend


def workload_text(scene, wl, W, H, spp, bounces, filt, lattice_dims=None):
    if lattice_dims:
        return (f"{wl.info.num_triangles} triangles: suzanne.obj on a jittered {lattice_dims} lattice (PCG32 seed 0x5EED), "
                f"{W}x{H} {spp}spp maxBounces {bounces}, {filt} filter iterations, MAX_FOLLOWS lifted "
                f"(BASELINE.json configs[4], SURVEY 8(d) config #5)")
    cam = "the reference's camera (main.cpp:42-51)" if scene == "room" else "SURVEY 8(d) auto camera + harness quad light"
    return f"{scene}.obj {W}x{H} {spp}spp maxBounces {bounces} + {filt}-iteration a-trous filter, {cam}"


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
    ap.add_argument("--lattice", default="37x37x38", help="copies of suzanne.obj for --scene lattice / cases.lattice")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = --size is the frame, split over the ranks (default: BASELINE's fixed frame); "
                         "weak = one --size tile per rank, the frame grows with N")
    ap.add_argument("--cases", default=None,
                    help="extra cases in the same JSON line: comma list of room, lattice, config4, or none "
                         "(default: room,lattice at N = 1 with the default scene; config4 at N > 1)")
    ap.add_argument("--rl", action="store_true",
                    help="run the Q-table guided kernel (reference trace3.cs traceRL, dead in its main) instead of trace(): "
                         "single GPU, no roofline / cpu_baseline (the table is shared mutable state)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    if a.steps < 1:
        sys.exit("--steps must be at least 1")
    lattice = a.scene == "lattice"
    default_run = (a.scene == "scene" and a.size is None and a.spp is None and a.bounces is None and a.filter is None)
    # defaults: the metric configuration; for the lattice, config #5's
    a.size = a.size or ("3840x2160" if lattice else "1920x1080")
    a.spp = a.spp if a.spp is not None else (1 if lattice else 4)
    a.bounces = a.bounces if a.bounces is not None else (4 if lattice else 5)
    a.filter = a.filter if a.filter is not None else (0 if lattice else 5)
    max_follows = (1 << 30) if lattice else 1000
    W, H = (int(v) for v in a.size.lower().split("x"))

    import torch
    from raytracer_amd import hiplib, parallel, workloads
    K = hiplib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit(f"--gpus {a.gpus} needs a torch.distributed launch with {a.gpus} ranks")
        a.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the hot path has no CPU fallback")
    # RT_BENCH_COLLECTIVES=gloo is a rehearsal of the N > 1 bookkeeping on a one-GPU box: every rank uses device 0 and
    # the tiles travel through host memory (RCCL refuses two ranks on one GPU).  The driver's runs use the default:
    # one GPU per rank, the launcher's own RCCL gather.
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
    cases_wanted = (a.cases.split(",") if a.cases not in (None, "none", "") else
                    ([] if a.cases is not None else
                     (["room", "lattice"] if world == 1 and default_run else (["config4"] if world > 1 and default_run else []))))

    ctx = hiplib.Context(local_rank)

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # how the tiles of the N-rank frame reach rank 0: "local" (N = 1), "rccl" (rt_gather_tiles, the product path),
    # "gloo-fallback" (rt_comm_init failed on some rank), "gloo-rehearsal" (RT_BENCH_COLLECTIVES=gloo)
    gather_state = {"mode": "local" if world == 1 else ("gloo-rehearsal" if rehearsal else "rccl"), "errors": None}

    def tiled_run(wl, W, H, spp, bounces, filt, steps, warmup, mf):
        """K timed frames of the tiled N-rank path (N = 1: one tile, the gather is a local copy when a communicator
        exists).  Returns (elapsed max over ranks, per-rank counters / timings, frame, backend)."""
        ctx.upload_scene(wl.buffers)
        ctx.resize(W, H)
        backend = parallel.HipBackend(ctx, wl, spp, bounces, filt, device, max_follows=mf)
        gather_group = dist
        if world > 1 and not rehearsal:
            # The launcher's own RCCL gather.  Should a rank fail to join (never seen; 8-GPU nodes are not ours to
            # test on), every rank falls back to the blocking torch.distributed gather through host memory (a gloo
            # group) rather than losing the run: same frame, slower exchange, and the JSON line says so.
            ok, why = 1.0, ""
            try:
                backend.init_comm(rank, world, dist)
            except Exception as e:  # noqa: BLE001
                why = f"{type(e).__name__}: {e}"
                ok = 0.0
            agree = torch.tensor([ok], device=device)
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)
            if float(agree[0]) == 0.0:
                # LOUD: every rank's RCCL error on stderr, and the JSON line says "gather": "gloo-fallback" at top level
                reasons = [None] * world
                dist.all_gather_object(reasons, why or "joined (another rank did not)")
                if rank == 0:
                    print("[bench] WARNING: the launcher's RCCL communicator (rt_comm_init) could not be set up; the tiles are "
                          "gathered through host memory over gloo instead — this is NOT the product path:", file=sys.stderr)
                    for r, text in enumerate(reasons):
                        print(f"[bench]   rank {r}: {text}", file=sys.stderr)
                    sys.stderr.flush()
                gather_state["mode"] = "gloo-fallback"
                gather_state["errors"] = reasons
                if backend.comm_ready:
                    ctx.comm_destroy()
                    backend.comm_ready = False
                gather_group = _GlooGather(dist)
            else:
                gather_state["mode"] = "rccl"
        frame = parallel.TiledFrame(backend, W, H, rank, world, hiplib.IMAGE_FILTERED if filt else hiplib.IMAGE_COLOR)

        def step():
            frame.render(0)
            frame.gather(gather_group)

        settle = 0
        while settle < 6:  # set-up: let the launcher settle its schedule (it times two candidates on the first frames)
            undecided = torch.tensor([1.0 if ctx.walk_stop() == 0 else 0.0])
            if dist is not None:
                if not rehearsal:
                    undecided = undecided.to(device)
                dist.all_reduce(undecided, op=dist.ReduceOp.MAX)
            if float(undecided[0]) == 0.0:
                break
            step()
            settle += 1
        for _ in range(warmup):
            step()
        fence()
        ctx.reset_counters()
        ctx.profile(True)
        ctx.timings(reset=True)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
        return elapsed, ctx.counters(), ctx.timings(), frame, backend, settle, step

    # ------------------------------------------------------------------ the headline (driver-timed) measurement
    tileW, tileH = W, H
    if world > 1 and a.scaling == "weak":
        halo = hiplib.lib().rt_filter_halo(a.filter) if a.filter > 0 else 0
        gx, gy = parallel.tile_grid(W, H, world, halo)
        W, H = W * gx, H * gy
    if a.rl:  # ---------------------------------------------- the Q-table guided kernel: a measurement of its own
        if world != 1 or lattice:
            sys.exit("--rl: single GPU, OBJ scenes only (the Q-table is shared mutable state and does not shard)")
        wl = workloads.Workload(a.scene, W, H, with_qtable=True)
        ctx.upload_scene(wl.buffers)
        ctx.resize(W, H)
        b = parallel.HipBackend(ctx, wl, a.spp, a.bounces, a.filter, device, kernel=hiplib.KERNEL_TRACE3_RL)
        for _ in range(a.warmup):
            b.trace((0, 0, W, H), 0)
        ctx.sync()
        ctx.reset_counters()
        ctx.profile(True)
        ctx.timings(reset=True)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            b.trace((0, 0, W, H), 0)
            b.filter((0, 0, W, H))
        ctx.sync()
        el = time.perf_counter() - t0
        cnt, tm = ctx.counters(), ctx.timings()
        visits = ctx.download(10, np.int32, len(wl.buffers[10]) // 4)
        print(json.dumps({"metric": f"Mrays/s (traceRL+filter), {a.scene}.obj {W}x{H} {a.spp}spp", "value": round(cnt.rays / el / 1e6, 3),
                          "unit": "Mrays/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": round(el * 1e3 / a.steps, 4), "higher_is_better": True, "scaling": "strong",
                          "vs_baseline": None, "dtype": "f32", "data": f"the reference's own {a.scene}.obj + its Q-table initialisation",
                          "config": {"workload": f"{a.scene}.obj {W}x{H} {a.spp}spp maxBounces {a.bounces}, RT_KERNEL_TRACE3_RL "
                                                 f"(Q-table guided, reference trace3.cs:695-791), {a.filter} filter iterations"},
                          "rays_per_frame": cnt.rays / a.steps, "q_updates_total": int(visits.sum()),
                          "kernel_ms_per_frame": {K.TIME_NAMES[i]: round(tm.ms[i] / a.steps, 4) for i in range(len(K.TIME_NAMES)) if tm.launches[i]}}))
        return
    if lattice:
        nx, ny, nz = (int(v) for v in a.lattice.lower().split("x"))
        wl = workloads.LatticeWorkload(W, H, nx, ny, nz)
    else:
        wl = workloads.Workload(a.scene, W, H)
    scene_key = wl.name if lattice else a.scene
    hit = stop_at_root = None
    if world == 1:
        ctx.upload_scene(wl.buffers)
        ctx.resize(W, H)
        hit, stop_at_root = frame_statistics(K, ctx, parallel.HipBackend(ctx, wl, a.spp, a.bounces, a.filter, device,
                                                                       max_follows=max_follows), wl, W, H, a.spp)
    elapsed, cnt, tm, frame, backend, settle, step = tiled_run(wl, W, H, a.spp, a.bounces, a.filter, a.steps, a.warmup, max_follows)
    ctx.profile(False)
    in_flight = ctx.frames_in_flight()
    alone = alone_roofline(K, ctx, step, a.steps, wl, W, H, a.spp, a.filter, scene_key, stop_at_root or 0)
    latency, alone_b2b = one_frame_latency(ctx, step, max(3, min(10, a.steps)))
    steady = None
    if world == 1 and not os.environ.get("RT_PIPELINE"):
        try:
            steady = steady_state(ctx, step)
        except Exception as e:  # noqa: BLE001  (an extra: never fatal for the measurement)
            steady = {"error": f"{type(e).__name__}: {e}"}
    if dist is not None:  # the slowest rank's
        lat = torch.tensor([latency, alone_b2b], dtype=torch.float64, device="cpu" if rehearsal else device)
        dist.all_reduce(lat, op=dist.ReduceOp.MAX)
        latency, alone_b2b = round(float(lat[0]), 4), round(float(lat[1]), 4)

    frame_sha = None
    if os.environ.get("RT_BENCH_CHECKSUM") and rank == 0:  # rehearsal aid: every N must assemble the same picture
        import hashlib
        frame_sha = hashlib.sha256(np.ascontiguousarray(frame.result()).tobytes()).hexdigest()
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
    walk_stop = ctx.walk_stop()
    # Opt-in extra, outside the measured value: the same frames with RT_FLAG_FAST_SHADOWS (not the reference's
    # algorithm: shadow rays that miss their light are not cast, the others stop at the first nearer hit).
    fast_ms = None
    if world == 1:
        ctx.profile(False)
        backend.flags = hiplib.FLAG_FAST_SHADOWS
        frame.render(0)
        fence()
        t1 = time.perf_counter()
        for _ in range(max(1, a.steps // 2)):
            frame.render(0)
        fence()
        fast_ms = (time.perf_counter() - t1) * 1e3 / max(1, a.steps // 2)
        backend.flags = 0
    local = torch.tensor([elapsed, float(cnt.rays), float(cnt.node_visits), float(cnt.tri_tests),
                          float(cnt.shaded_hits), float(cnt.paths)] + [tm.ms[i] for i in range(len(K.TIME_NAMES))] + [frame_rays],
                         dtype=torch.float64, device="cpu" if (rehearsal or dist is None) else device)
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

    # ------------------------------------------------------------------ N > 1: BASELINE configs[3] strong-scaled
    cases = {}
    if "config4" in cases_wanted and world > 1:
        try:
            W4, H4, spp4 = 3840, 2160, 16
            wl4 = workloads.Workload("scene", W4, H4)
            e4, c4, t4, f4, b4, _, _ = tiled_run(wl4, W4, H4, spp4, 5, 5, max(2, a.steps // 4), 1, 1000)
            ctx.profile(False)
            ctx.reset_counters()
            b4.trace(f4.tile, 0)
            ctx.sync()
            l4 = torch.tensor([e4, float(ctx.counters().rays)], dtype=torch.float64,
                              device="cpu" if rehearsal else device)
            m4, s4 = l4.clone(), l4.clone()
            dist.all_reduce(m4, op=dist.ReduceOp.MAX)
            dist.all_reduce(s4, op=dist.ReduceOp.SUM)
            n4 = max(2, a.steps // 4)
            cases["config4"] = {"workload": workload_text("scene", wl4, W4, H4, spp4, 5, 5) + " (BASELINE.json configs[3])",
                                "scaling": "strong", "n_gpus": world, "steps": n4,
                                "ms_per_frame": round(float(m4[0]) * 1e3 / n4, 4),
                                "value": round(float(s4[1]) * n4 / float(m4[0]) / 1e6, 3), "unit": "Mrays/s",
                                "rays_per_frame": float(s4[1]),
                                "parallelism": f"{f4.grid[0]} x {f4.grid[1]} tiles, halo {b4.halo} px"}
        except Exception as e:
            cases["config4"] = {"error": f"{type(e).__name__}: {e}"}
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    rays, visits, tests, hits, paths = sums[1:6]
    frame_rays = sums[6 + len(K.TIME_NAMES)]
    ms_per_step = elapsed * 1e3 / a.steps
    mrays = frame_rays * a.steps / elapsed / 1e6
    rl = with_alone(roofline(K, tm, cnt, wl, W, H, a.spp, a.filter, a.steps, scene_key, stop_at_root or 0), alone)
    if rl is not None and world == 1:
        # the whole timed region against the same roof: every traversal's algorithmic operations / the driver-timed interval
        ops = 25.0 * float(cnt.node_visits) + 51.0 * float(cnt.tri_tests)
        rl["timed_region"] = {"algorithmic_ops_per_frame": round(ops / a.steps), "achieved_Tops": round(ops / elapsed / 1e12, 4),
                              "frac_of_valu_peak": round(ops / elapsed / 1e12 / VALU_PEAK_TOPS_NOFMA, 4),
                              "note": "all kernels of the frame, frames in flight included: trace + filter time, traversal operations only"}
    frame_name = f"{scene_key}{'.obj' if not lattice else ''} {W}x{H} {a.spp}spp"
    headline = (a.scene, W, H, a.spp, a.bounces, a.filter) == ("scene", 1920, 1080, 4, 5, 5)
    out = {
        "metric": "Mrays/s (trace+filter), scene.obj 1080p 4spp" if headline else f"Mrays/s (trace+filter), {frame_name}",
        "value": round(mrays, 3), "unit": "Mrays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 4), "ms_per_frame": round(ms_per_step, 4),
        "ms_per_step_is": THROUGHPUT_NOTE, "frames_in_flight": in_flight,
        "latency_ms_one_frame": latency, "ms_per_frame_one_frame_at_a_time": alone_b2b,
        "gather": gather_state["mode"], **({"gather_errors": gather_state["errors"]} if gather_state["errors"] else {}),
        "higher_is_better": True, "scaling": a.scaling if world > 1 else "strong", "vs_baseline": None, "dtype": "f32",
        "data": ("the reference's own res/scene.obj (fixture copy) + harness quad light; no external data" if a.scene == "scene"
                 else "synthetic lattice of the reference's res/suzanne.obj" if lattice else f"the reference's own {a.scene}.obj"),
        **({"rehearsal": "gloo collectives, all ranks on GPU 0: not a measurement"} if rehearsal else {}),
        "config": {"workload": workload_text(a.scene, wl, W, H, a.spp, a.bounces, a.filter, a.lattice if lattice else None)
                               + (" (BASELINE.json configs[2])" if headline else ""),
                   "triangles": wl.info.num_triangles, "nodes": wl.info.num_nodes, "emitters": wl.num_emitters,
                   "schedule": "wavefront pipeline, samples of a frame traced concurrently (rt_trace_batch)",
                   "frame": f"{W}x{H}",
                   "parallelism": f"{world} ranks as a {frame.grid[0]} x {frame.grid[1]} grid of screen-space tiles"
                                  + (f" of {tileW}x{tileH} (the frame grows with N)" if world > 1 and a.scaling == "weak" else "")
                                  + f", halo {backend.halo} px, one gather to rank 0"
                                  + (" (rt_gather_tiles: RCCL send/recv on a second stream, event-ordered)" if frame.native else
                                     " (torch.distributed gather through host memory)" if world > 1 else "")},
        "rays_per_frame": frame_rays, "rays_traced_per_frame_incl_halo": rays / a.steps,
        "node_visits_per_frame": visits / a.steps,
        "tri_tests_per_frame": tests / a.steps,
        "walk_stop": walk_stop, "schedule_settle_frames": settle,
        "kernel_ms_per_frame_max_rank": {K.TIME_NAMES[i]: round(maxs[6 + i] / a.steps, 4) for i in range(len(K.TIME_NAMES))
                                         if maxs[6 + i] > 0},
        "roofline": rl,
    }
    if hit is not None:
        out["hit_pixel_fraction"] = round(hit, 5)
        out["rays_past_root_per_frame"] = frame_rays - float(stop_at_root)
        out["rays_per_frame_if_primary_per_sample"] = frame_rays + float((a.spp - 1) * W * H)
    if steady is not None:
        out["steady_state"] = steady
    if frame_sha is not None:
        out["frame_sha256"] = frame_sha
    if fast_ms is not None:
        out["opt_in"] = {"fast_shadows_ms_per_frame": round(fast_ms, 4),
                         "note": "RT_FLAG_FAST_SHADOWS: light-first any-hit shadow rays (SURVEY 8(f) rank 4); same image up "
                                 "to exact-distance ties, fewer rays traced; not part of `value`"}
    if world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(wl, a.spp, a.bounces, a.filter, max_follows, 15.0 if cases_wanted else 25.0)
        except Exception as e:  # the checker library is test infrastructure; never fatal for the measurement
            out["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "port",
                                   "sample": f"unavailable: {e}"}
    # ------------------------------------------------------------------ N == 1: the throughput-bound cases
    if world == 1:
        budget = 0.0 if a.no_cpu_baseline else 8.0
        if "room" in cases_wanted:
            try:
                wlr = workloads.Workload("room", 1920, 1080)
                c, _ = run_case(K, parallel, ctx, device, wlr, "room", 1920, 1080, 4, 5, 5, 1000, max(3, a.steps // 2), 2, budget)
                c["workload"] = workload_text("room", wlr, 1920, 1080, 4, 5, 5)
                cases["room"] = c
            except Exception as e:
                cases["room"] = {"error": f"{type(e).__name__}: {e}"}
        if "lattice" in cases_wanted:
            try:
                nx, ny, nz = (int(v) for v in a.lattice.lower().split("x"))
                wll = workloads.LatticeWorkload(3840, 2160, nx, ny, nz)
                c, _ = run_case(K, parallel, ctx, device, wll, wll.name, 3840, 2160, 1, 4, 0, 1 << 30, max(3, a.steps // 4), 2, budget)
                c["workload"] = workload_text("lattice", wll, 3840, 2160, 1, 4, 0, a.lattice)
                cases["lattice"] = c
            except Exception as e:
                cases["lattice"] = {"error": f"{type(e).__name__}: {e}"}
    if cases:
        out["cases"] = cases
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
