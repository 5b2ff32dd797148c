#!/bin/bash
# HBM-side bytes of the a-trous passes on a 1080p frame of scene $1 (default room): separate rocprofv3 --pmc passes for FETCH_SIZE
# and WRITE_SIZE over scripts/filter_time.py, one frame at a time.  Output: gpurun_out/filter_traffic_$1[_$TAG].txt
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SCENE=${1:-room}; TAG=${2:-}
export RT_PIPELINE=1
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/filter_traffic_${SCENE}${TAG:+_$TAG}.txt
: > $OUT
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/ft_$ctr
  timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d /tmp/ft_$ctr -o ft -- python3 $REPO/scripts/filter_time.py $SCENE > /tmp/ft_$ctr.log 2>&1 || { tail -5 /tmp/ft_$ctr.log; exit 1; }
  f=$(find /tmp/ft_$ctr -name "*counter_collection.csv" | head -1); [ -n "$f" ] || { tail -20 /tmp/ft_$ctr.log; exit 1; }
  python3 - "$f" $ctr >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "atrous" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]:
        acc[(r["Kernel_Name"][:40], r.get("LDS_Block_Size", "?"))].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    tail = v[len(v) // 2:]  # the frames after the trace (the first half runs on a cleared image)
    print(sys.argv[2], k, "launches", len(v), "avg_KB_all", round(sum(v) / len(v), 1), "avg_KB_second_half", round(sum(tail) / len(tail), 1))
PY
done
cat $OUT
