#!/bin/bash
# HBM traffic of the wavefront kernels on room.obj 1080p (one PMC group per run).
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_room_mem_${1:-x}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --scene room --steps 2 --warmup 3 --no-cpu-baseline"
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-30)
  timeout -s KILL 240 rocprofv3 --pmc $grp --output-format csv -d "$OUT/$name" -- $BENCH > "$OUT/$name.log" 2>&1 || { echo "pmc run $grp failed"; tail -5 "$OUT/$name.log"; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][-28:], r["Counter_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    if "rt_" in k:
        print(f"{k:30s} {c:32s} per dispatch {v / n:.4g} ({n} dispatches)")
PY
