"""Dev tool: per-launch table of `k_search` inside the inference frame from a rocprofv3 kernel trace of an EAGER, one-frame-in-
flight bench run (`--no-graph --in-flight 1`): for every position in the frame (the occupancy probe launch, then the wave-front
iterations 0, 1, ...) the workgroups, workgroups / 1024 resident slots (256 CUs x 4 workgroups of 256 threads at 108 VGPRs) and
the mean / min duration.  VERDICT r05 item 4: do launches land just above an integer number of resident rounds?

    python tools/search_launches.py <dir>/r_kernel_trace.csv [frames_to_skip]
"""
import collections
import csv
import sys


def main(path, skip=3):
    rows = []
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "k_search" not in name and "k_probe_points" not in name:
            continue
        wg = int(r["Workgroup_Size_X"]) if "Workgroup_Size_X" in r else int(r["Workgroup_Size"])
        grid = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "probe_points" if "k_probe_points" in name else "search", grid // wg))
    rows.sort()
    frames, cur = [], None
    for s, e, kind, nwg in rows:
        if kind == "probe_points":          # a frame's first launch on this path
            cur = []
            frames.append(cur)
        elif cur is not None:
            cur.append((nwg, (e - s) / 1e3))
    frames = [f for f in frames[skip:] if f]
    if not frames:
        print("no frames found")
        return
    n_pos = max(len(f) for f in frames)
    print("frames analysed: %d (skipped %d); launches per frame: %s" % (len(frames), skip, sorted(collections.Counter(len(f) for f in frames).items())))
    print("%-10s %10s %12s %10s %10s %10s" % ("position", "workgroups", "wg/1024", "mean us", "min us", "us/round"))
    tot = 0.0
    for p in range(n_pos):
        v = [f[p] for f in frames if len(f) > p]
        wg = sum(x[0] for x in v) / len(v)
        us = sum(x[1] for x in v) / len(v)
        tot += us * len(v) / len(frames)
        rounds = -(-int(wg) // 1024)
        print("%-10s %10.0f %12.2f %10.1f %10.1f %10.1f" % ("probe" if p == 0 else "iter %d" % (p - 1), wg, wg / 1024.0, us, min(x[1] for x in v), us / max(rounds, 1)))
    print("k_search per frame: %.1f us" % tot)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3)
