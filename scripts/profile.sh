#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box:  scripts/profile.sh r01
# kernel-trace stats in one run; each PMC group in a run of its own (never combined with traces).
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --cases none ${BENCH_ARGS:-}"   # (bench.py first lets the launcher settle its schedule: no WalkTune frames in the counters)
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1 || { echo "kernel-trace run failed"; tail -5 "$OUT/trace.log"; exit 1; }
# from here on one frame at a time: a launch's duration and counters are then its own (with frames in flight they include
# what the other frames' kernels do meanwhile)
export RT_PIPELINE=1
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace1" -- $BENCH > "$OUT/trace1.log" 2>&1 || { echo "kernel-trace run (one frame at a time) failed"; tail -5 "$OUT/trace1.log"; exit 1; }
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 240 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$name" -- $BENCH > "$OUT/pmc_$name.log" 2>&1 || { echo "pmc run $grp failed"; tail -5 "$OUT/pmc_$name.log"; exit 1; }
done
find "$OUT" -name "*.csv" | head -40
