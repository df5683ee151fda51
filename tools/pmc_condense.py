"""Condenses a raw `rocprofv3 --pmc ... --kernel-trace` pass (r_counter_collection.csv, tens of MB) into one row per
(kernel, counter): sum over launches, launches, total duration -- the form committed under profiles/r02_pmc/.

    python tools/pmc_condense.py gpurun_out/pmc2_<set>/ profiles/r02_pmc/pmc2_<set>.csv
"""
import collections
import csv
import sys


def main(src, dst):
    acc = collections.defaultdict(lambda: [0.0, set(), 0.0])
    for r in csv.DictReader(open(src.rstrip("/") + "/r_counter_collection.csv")):
        k = (r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])
        a = acc[k]
        a[0] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in a[1]:
            a[1].add(r["Dispatch_Id"])
            a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    w = csv.writer(open(dst, "w"))
    w.writerow(["kernel", "counter", "sum", "launches", "total_ns"])
    for (k, c), (v, ds, ns) in sorted(acc.items(), key=lambda x: -x[1][2]):
        w.writerow([k, c, "%.6g" % v, len(ds), "%.0f" % ns])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
