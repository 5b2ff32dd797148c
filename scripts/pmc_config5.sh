#!/bin/bash
# HBM traffic and L2 hit rate of the kernels on SURVEY 8(d) config #5 (one PMC group per run).
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_config5_${1:-x}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/scripts/config5.py --reps 1 --crop 8 --out $OUT/config5.json"
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-30)
  timeout -s KILL 240 rocprofv3 --pmc $grp --output-format csv -d "$OUT/$name" -- $CMD > "$OUT/$name.log" 2>&1 || { echo "pmc run $grp failed"; tail -5 "$OUT/$name.log"; exit 1; }
  echo "done $grp"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").split("<")[0], r["Counter_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    if "rt_" in k:
        print(f"{k:30s} {c:32s} total {v:.4g} over {n} dispatches")
PY
