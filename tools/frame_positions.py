"""Dev tool (round 6): per-launch-POSITION durations of the inference frame from a rocprofv3 kernel trace of a one-frame-in-flight run
(tools/prof_frame.sh writes gpurun_out/prof_frame/r_kernel_trace.csv).  Frames are split at k_smpl_tfs; only frames with the most
common launch count are averaged.   usage: python tools/frame_positions.py [kernel_trace.csv] [substring ...]"""
import collections
import csv
import statistics
import sys

path = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".csv") else "gpurun_out/prof_frame/r_kernel_trace.csv"
only = [a for a in sys.argv[1:] if not a.endswith(".csv")]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
frames, cur = [], []
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if name.startswith("k_smpl_tfs"):
        if cur:
            frames.append(cur)
        cur = []
    cur.append((name, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
frames = frames[len(frames) // 5:len(frames) - len(frames) // 10]          # drop warm-up and tail
L = collections.Counter(len(f) for f in frames).most_common(1)[0][0]
fs = [f for f in frames if len(f) == L]
print("frames averaged: %d of %d, launches per frame: %d" % (len(fs), len(frames), L))
total = 0.0
for i in range(L):
    m = statistics.mean(f[i][1] for f in fs)
    gap = statistics.mean((f[i][2] - f[i - 1][3]) / 1e3 for f in fs) if i else 0.0
    total += m
    if not only or any(o in fs[0][i][0] for o in only):
        print("%3d %-44s %8.1f us   gap before %6.1f" % (i, fs[0][i][0][:44], m, gap))
print("sum of kernel time per frame %.1f us, frame span %.1f us" % (total, statistics.mean((f[-1][3] - f[0][2]) / 1e3 for f in fs)))
