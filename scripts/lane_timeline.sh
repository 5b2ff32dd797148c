#!/bin/bash
# kernel trace of 40 metric frames in flight -> scripts/lane_timeline.py
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_lanes_${1:-x}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 $REPO/bench.py --steps 40 --warmup 5 --no-cpu-baseline --cases none > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
python3 $REPO/scripts/lane_timeline.py $(find "$OUT/trace" -name "*kernel_trace.csv") 4 9
