#!/bin/bash
# Is the extend kernel bound by the vector L1 (TA / TCP / TD) or by VALU issue?  Busy and stall cycles of the texture path
# beside the SQ's, one PMC group per run, one frame at a time (RT_PIPELINE=1).  $1: tag, $2: bench.py --scene value (room).
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_l1_${1:-x}
SCENE=${2:-room}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export RT_PIPELINE=1
BENCH="python3 $REPO/bench.py --scene $SCENE --steps 3 --warmup 1 --no-cpu-baseline --cases none"
i=0
for grp in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE" \
           "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_sum SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/run$i" -- $BENCH > "$OUT/run$i.log" 2>&1 || { echo "pmc run $grp failed"; tail -5 "$OUT/run$i.log"; exit 1; }
  echo "done $grp"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").split("<")[0], r["Counter_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
out = open(sys.argv[1] + "/summary.csv", "w")
out.write("kernel,counter,per_dispatch,dispatches\n")
for (k, c), (v, n) in sorted(agg.items()):
    if "rt_wf" in k:
        out.write(f"{k},{c},{v / n:.6g},{n}\n")
out.close()
print(open(sys.argv[1] + "/summary.csv").read())
PY
