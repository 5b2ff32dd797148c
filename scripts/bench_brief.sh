#!/bin/bash
# bench.py + a short summary of the numbers that are compared from build to build.  $1 = tag of the output files.
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p gpurun_out
TAG=${1:-x}; shift
timeout -k 10 800 python bench.py --steps 20 --warmup 5 "$@" > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { tail -c 1500 gpurun_out/bench_$TAG.err; exit 1; }
python - "$TAG" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/bench_{sys.argv[1]}.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "frames_in_flight", "latency_ms_one_frame", "ms_per_frame_one_frame_at_a_time")})
print(d.get("kernel_ms_per_frame_max_rank"), "steady", {k: (d.get("steady_state") or {}).get(k) for k in ("frames_in_flight", "ms_per_frame", "error")})
r = d.get("roofline") or {}
print("roofline", {k: r.get(k) for k in ("kernel", "frac", "avg_launch_ms")}, "alone", (r.get("one_frame_at_a_time") or {}))
for k, c in (d.get("cases") or {}).items():
    rr = c.get("roofline") or {}
    print(k, {a: c.get(a) for a in ("ms_per_frame", "latency_ms_one_frame", "ms_per_frame_one_frame_at_a_time", "value", "error")},
          c.get("kernel_ms_per_frame"), {a: rr.get(a) for a in ("kernel", "frac")}, (rr.get("one_frame_at_a_time") or {}).get("frac"))
PY
