"""Per-lane timeline of the metric frame with frames in flight, from a rocprofv3 --kernel-trace CSV: for the frames in
the middle of the run, how long each launch of a frame's chain lasts and how long the lane's stream is empty before it.
usage: python scripts/lane_timeline.py <kernel_trace.csv> [first_frame last_frame]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
by_stream = collections.defaultdict(list)
for r in rows:
    by_stream[r["Stream_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:44]


lanes = {s: sorted(v) for s, v in by_stream.items() if sum("rt_wf_blend" in k[2] for k in v) >= 8}
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3, 8)
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
chains = []
for s, v in sorted(lanes.items()):
    # a frame on a lane = from one classify launch to the launch before the next one
    starts = [i for i, k in enumerate(v) if "rt_wf_classify" in k[2]]
    for f in range(lo, min(hi, len(starts) - 1)):
        a, b = starts[f], starts[f + 1]
        a0 = a
        while a0 > 0 and "rt_wf_classify" not in v[a0 - 1][2] and "atrous" not in v[a0 - 1][2]:
            a0 -= 1   # (memsets / fills queued before the classify launch belong to the frame)
        frame = v[a0:b]
        prev_end = v[a0 - 1][1] if a0 > 0 else frame[0][0]
        chains.append((frame[-1][1] - prev_end) / 1e3)
        pos = collections.Counter()
        for st, en, name in frame:
            key = short(name)
            idx = pos[key]
            pos[key] += 1
            g = agg[(key, idx)]
            g[0] += 1
            g[1] += (en - st) / 1e3
            g[2] += max(0, st - prev_end) / 1e3
            prev_end = en
print(f"{len(lanes)} lanes, frames {lo}..{hi - 1} of each; lane time per frame (end of previous frame's last launch -> end of this one's): "
      f"mean {sum(chains) / len(chains):.1f} us over {len(chains)}")
tot_d = tot_g = 0.0
order = []
for (key, idx), (n, d, g) in agg.items():
    order.append((key, idx, n, d / n, g / n))
for key, idx, n, d, g in order:
    print(f"{key:46s} #{idx}  n {n:3d}  duration {d:8.1f} us  empty before {g:7.1f} us")
    tot_d += d
    tot_g += g
print(f"sum of durations {tot_d:.1f} us, sum of gaps {tot_g:.1f} us")
