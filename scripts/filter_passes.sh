#!/bin/bash
# Per-pass kernel durations of the a-trous filter on the metric frame, with and without the flat-cell record
# (rocprofv3 kernel trace of scripts/filter_time.py).  Output: gpurun_out/filter_passes_{on,off}.txt
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for mode in on off; do
  if [ $mode = off ]; then export RT_FILTER_CELLS=0; else unset RT_FILTER_CELLS; fi
  rm -rf /tmp/fp_$mode
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp_$mode -o fp -- python3 $REPO/scripts/filter_time.py > /tmp/fp_$mode.log 2>&1 || { tail -5 /tmp/fp_$mode.log; exit 1; }
  f=$(find /tmp/fp_$mode -name "*kernel_stats.csv" | head -1); [ -n "$f" ] || { tail -20 /tmp/fp_$mode.log; find /tmp/fp_$mode | head; exit 1; }
  python3 - "$f" > $REPO/gpurun_out/filter_passes_$mode.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "atrous" in n or "flat" in n or "blend" in n:
        print(n[:70], r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 2), "min_us", round(float(r["MinNs"]) / 1e3, 2))
PY
  t=$(find /tmp/fp_$mode -name "*kernel_trace.csv" | head -1)
  python3 - "$t" >> $REPO/gpurun_out/filter_passes_$mode.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "atrous" in r["Kernel_Name"] or "flat_levels" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print("last frame, in order:")
for r in rows[-6:]:
    print("  ", r["Kernel_Name"][:48], "grid", r.get("Grid_Size_X", r.get("Grid_Size", "?")), "lds", r.get("LDS_Block_Size", "?"), "us", round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 2))
PY
done
