#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (scripts/profile.sh) into the small files kept under profiles/:
  profiles/<name>_kernel_stats.csv   rocprofv3 --kernel-trace --stats: calls, total / average / min / max ns per kernel
  profiles/<name>_pmc_summary.csv    average per dispatch of every counter collected, per kernel
and prints the HBM bytes per launch of each kernel (2 x FETCH_SIZE + WRITE_SIZE, in 64-byte units on gfx950: see
MI355X_MICROARCH.md) for profiles/traffic.json.
usage: python scripts/summarize_profile.py r01c r01_final"""
import collections, csv, glob, json, os, sys

tag, name = sys.argv[1], sys.argv[2]


def kname(full):
    """'void rt_wf_extend_kernel<2>(RtScene, ...)' -> 'rt_wf_extend_kernel' (template variants are summed)."""
    k = full.replace("(anonymous namespace)::", "").split("(")[0].strip()
    if k.startswith("void "):
        k = k[5:]
    return k.split("<")[0]

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")

stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(dst, f"{name}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "average_ns", "percentage", "min_ns", "max_ns"])
        for r in rows:
            w.writerow([kname(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    print("kernel stats:", len(rows), "kernels")

stats1 = glob.glob(os.path.join(src, "trace1", "**", "*kernel_stats.csv"), recursive=True)
if stats1:  # RT_PIPELINE=1: one frame at a time
    rows = list(csv.DictReader(open(stats1[0])))
    with open(os.path.join(dst, f"{name}_kernel_stats_one_frame_at_a_time.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "average_ns", "percentage", "min_ns", "max_ns"])
        for r in rows:
            w.writerow([kname(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = (kname(r["Kernel_Name"]), r["Counter_Name"])
        agg[k][0] += float(r["Counter_Value"])
        agg[k][1] += 1
with open(os.path.join(dst, f"{name}_pmc_summary.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "avg_per_dispatch", "dispatches"])
    for (k, c), (v, n) in sorted(agg.items()):
        if k.startswith("rt_"):
            w.writerow([k, c, v / n, n])
traffic = {}
for (k, c), (v, n) in agg.items():
    if c == "FETCH_SIZE" and k.startswith("rt_"):
        wr = agg.get((k, "WRITE_SIZE"), [0.0, 1])
        # FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3's derived metric; gfx950 under-reports FETCH by 2x
        traffic[k] = round(2 * 1024 * v / n + 1024 * wr[0] / max(wr[1], 1))
print(json.dumps(traffic, indent=1))
