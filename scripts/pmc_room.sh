#!/bin/bash
# SQ counters of the wavefront kernels on room.obj 1080p (the case with the most secondary rays).
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_room_${1:-x}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --scene room --steps 2 --warmup 2 --no-cpu-baseline --cases none ${BENCH_ARGS:-}"   # BENCH_ARGS="--rl --spp 4 --filter 0": the Q-table kernel
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/sq" -- $BENCH > "$OUT/sq.log" 2>&1 || { echo failed; tail -5 "$OUT/sq.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM --output-format csv -d "$OUT/sq2" -- $BENCH > "$OUT/sq2.log" 2>&1 || { echo failed2; tail -5 "$OUT/sq2.log"; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM --output-format csv -d "$OUT/sq3" -- $BENCH > "$OUT/sq3.log" 2>&1 || { echo failed3; tail -5 "$OUT/sq3.log"; }
timeout -s KILL 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d "$OUT/sq4" -- $BENCH > "$OUT/sq4.log" 2>&1 || { echo failed4; tail -5 "$OUT/sq4.log"; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for d in ("sq", "sq2", "sq3", "sq4"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(sys.argv[1] + f"/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:28]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        print(k, {c: f"{x:.3g}" for c, x in v.items()})
        if "SQ_THREAD_CYCLES_VALU" in v:
            print("   lane utilisation", v["SQ_THREAD_CYCLES_VALU"] / (64 * v["SQ_INSTS_VALU"] + 1), " valu busy of wave cycles", 4 * v["SQ_ACTIVE_INST_VALU"] / (v["SQ_WAVE_CYCLES"] + 1))
PY
