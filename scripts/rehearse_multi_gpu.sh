#!/bin/bash
# Rehearsal of bench.py's N > 1 path on a ONE-GPU box: N ranks share GPU 0, collectives over gloo through host
# memory (RT_BENCH_COLLECTIVES=gloo).  Checks the code path and the bookkeeping (tiles, halo, gather: the assembled frame
# must be the one-rank frame bit for bit; frame-ray count), NOT performance.   usage: scripts/rehearse_multi_gpu.sh [ranks=2]
set -eu
N=${1:-2}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"
export RT_BENCH_COLLECTIVES=gloo RT_BENCH_CHECKSUM=1 MASTER_ADDR=127.0.0.1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --cases none > gpurun_out/rehearse_1.json
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus "$N" --steps 3 --warmup 1 --scaling strong --cases none > gpurun_out/rehearse_$N.json
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port 29518 \
    bench.py --gpus "$N" --steps 3 --warmup 1 --scaling weak --cases none > gpurun_out/rehearse_weak_$N.json
python - "$N" <<'PY'
import json, sys
n = sys.argv[1]
a = json.loads(open("gpurun_out/rehearse_1.json").read().strip().splitlines()[-1])
b = json.loads(open(f"gpurun_out/rehearse_{n}.json").read().strip().splitlines()[-1])
print("1 rank :", a["rays_per_frame"], a["ms_per_frame"])
print(f"{n} ranks:", b["rays_per_frame"], b["ms_per_frame"], "incl. halo", b["rays_traced_per_frame_incl_halo"], b["n_gpus"])
assert b["n_gpus"] == int(n) and b["rays_per_frame"] == a["rays_per_frame"], "the strips' rays must add up to the frame's"
assert b["rays_traced_per_frame_incl_halo"] >= b["rays_per_frame"]
assert b["frame_sha256"] == a["frame_sha256"], "the assembled frame differs from the one-rank frame"
print("assembled frame identical:", b["frame_sha256"][:16])
w = json.loads(open(f"gpurun_out/rehearse_weak_{n}.json").read().strip().splitlines()[-1])
print(f"{n} ranks, weak:", w["config"]["frame"], w["rays_per_frame"], w["scaling"], w["config"]["parallelism"])
assert w["scaling"] == "weak" and w["n_gpus"] == int(n) and w["rays_per_frame"] > a["rays_per_frame"]
print("ok")
PY
