#!/bin/bash
# Memory-side counters of config #5.  rocprofv3 on gfx950 offers no Infinity-Cache (MALL) hit counter
# (profiles/r03_rocprofv3_counter_names_gfx950.txt is the whole list); what the L2 does report about its misses:
#   TCC_EA0_RDREQ{,_32B,_64B,_128B}_sum  the read requests it sends to the fabric, by size  -> exact L2-miss bytes
#   TCC_EA0_RDREQ_LEVEL_sum / TCC_EA0_RDREQ_sum                                           -> average latency of such a read
# The latency is calibrated with random 64-byte gathers from a table that stays in the Infinity Cache (64 MiB) and one far
# beyond it (4 GiB), scripts/micro/ea_gather.hip.  One PMC group per run; RT_PIPELINE=1: one frame at a time, so that a
# launch's counters are its own.
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_memside_${1:-x}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export RT_PIPELINE=1
BENCH="python3 $REPO/bench.py --scene lattice --steps 2 --warmup 1 --no-cpu-baseline --cases none"
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_HIT_sum TCC_MISS_sum WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/bench$i" -- $BENCH > "$OUT/bench$i.log" 2>&1 || { echo "pmc run $grp failed"; tail -5 "$OUT/bench$i.log"; exit 1; }
  echo "done $grp"
done
for mib in 64 4096; do
  timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d "$OUT/gather$mib" -- $REPO/scripts/micro/ea_gather.bin $mib 256 > "$OUT/gather$mib.log" 2>&1 || { echo "gather $mib failed"; tail -5 "$OUT/gather$mib.log"; exit 1; }
  tail -1 "$OUT/gather$mib.log"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    tag = "gather" + f.split("/gather")[1].split("/")[0] + ":" if "/gather" in f else ""
    for r in csv.DictReader(open(f)):
        k = (tag + r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").split("<")[0], r["Counter_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
out = open(sys.argv[1] + "/summary.csv", "w")
out.write("kernel,counter,per_dispatch,dispatches\n")
for (k, c), (v, n) in sorted(agg.items()):
    if "rt_" in k or "ea_gather" in k:
        out.write(f"{k},{c},{v / n:.6g},{n}\n")
        print(f"{k:44s} {c:40s} per dispatch {v / n:.5g} ({n})")
PY
