"""Development aid: per-dispatch kernel durations of the last frames of a rocprofv3 --kernel-trace run of
scripts/stage_frames.py.  usage: python scripts/stage_times.py <dir with *_kernel_trace.csv> [label]"""
import csv, glob, sys
import numpy as np
d = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else d
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
seq = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]) for r in rows))
seq = [s for s in seq if s[2].startswith("rt_")]
starts = [i for i, s in enumerate(seq) if s[2] == "rt_wf_primary_kernel" or s[2] == "rt_wf_classify_kernel"]
frames = []
for a, b in zip(starts, starts[1:] + [len(seq)]):
    frames.append(seq[a:b])
frames = [fr for fr in frames if len(fr) == len(frames[-1])][-4:]   # the last (settled) frames
names = [s[2].replace("rt_wf_", "").replace("_kernel", "").replace("rt_", "") for s in frames[-1]]
dur = np.array([[(s[1] - s[0]) / 1e3 for s in fr] for fr in frames])
span = np.array([(fr[-1][1] - fr[0][0]) / 1e3 for fr in frames])
print(f"== {label}: frame span {span.mean():.0f} us (device, first start to last end), sum of kernels {dur.sum(1).mean():.0f} us")
print("   " + " ".join(f"{n}:{v:.0f}" for n, v in zip(names, dur.mean(0))))
tot = {}
for n, v in zip(names, dur.mean(0)):
    tot[n] = tot.get(n, 0) + v
print("   totals: " + " ".join(f"{n} {v:.0f}" for n, v in tot.items()))
