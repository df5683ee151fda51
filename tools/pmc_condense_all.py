"""ONE summary of every PMC pass of tools/pmc_all.sh, stamped with the hashes of the device code the counters were
collected on (bench.py refuses a summary whose hash differs from the library it runs):

  <round>_pmc_traffic.json  <round>_pmc_search.json  <round>_pmc_mfma.json   (bench passes, tools/pmc_summarise_bench.py's summaries)
  <round>_pmc_encode.json   isolated hash-grid lookup (tools/pmc_encode.py): requests per sample, hit rates, fabric bytes
  <round>_pmc_hgbwd.json    hash-grid backward: L2 atomic requests per launch against the measured ceiling

    python tools/pmc_condense_all.py <gpurun_out dir> <out dir>
"""
import collections
import csv
import glob
import hashlib
import json
import os
import statistics
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import pmc_summarise_bench as pmc_bench  # noqa: E402

SO = os.path.join(os.path.dirname(HERE), "instantavatar_amd", "libinstantavatar_hip.so")


ROUND = os.environ.get("IA_PMC_ROUND", "r05")


def so_hash():
    return hashlib.sha256(open(SO, "rb").read()).hexdigest()


def device_code():
    """{translation unit: hash of its gfx950 code objects} embedded in the library (instantavatar_amd/build.py)"""
    sys.path.insert(0, os.path.dirname(HERE))
    from instantavatar_amd import build as ia_build
    return ia_build.device_manifest(SO)


def per_kernel(pattern, match):
    """{kernel: {counter: per-launch mean}}, {kernel: median launch us} over the passes matching `pattern`"""
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    durs = collections.defaultdict(list)
    for f in sorted(glob.glob(pattern, recursive=True)):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            k = k[5:] if k.startswith("void ") else k
            if not any(m in k for m in match):
                continue
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
            if (k, r["Dispatch_Id"]) not in seen:
                seen.add((k, r["Dispatch_Id"]))
                durs[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    return {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in acc.items()}, {k: statistics.median(v) for k, v in durs.items()}, {k: len(v) for k, v in durs.items()}


def per_kernel_tail(pattern, match, last_n):
    """counters averaged over the LAST `last_n` dispatches of the kernels matching `match` (the measured launches of a
    script whose earlier launches of the same kernel belong to its set-up), and their median duration in us"""
    out, us = {}, []
    for f in sorted(glob.glob(pattern, recursive=True)):
        rows = [r for r in csv.DictReader(open(f)) if match in r["Kernel_Name"]]
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-last_n:]
        acc = collections.defaultdict(list)
        for r in rows:
            if int(r["Dispatch_Id"]) in ids:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                us.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
        for c, v in acc.items():
            out[c] = sum(v) / max(len(ids), 1)     # a counter may be reported in several rows per dispatch (per XCD): sum them
    return out, (statistics.median(us) if us else None), None


def main(root, out_dir):
    os.makedirs(out_dir, exist_ok=True)
    meta = {"so_sha256": so_hash(), "device_code": device_code(), "collected_by": "tools/pmc_all.sh (rocprofv3 --pmc <one set per pass> --kernel-trace)"}
    pmc_bench.main(root, out_dir, prefix=ROUND, meta=meta)
    # ---- isolated encoder
    C, us, _ = per_kernel(os.path.join(root, "pmc_enc_*", "**", "*counter_collection.csv"), ("k_hashgrid<", "k_encode_xcd"))
    V = 1 << 20
    enc = {}
    for k, c in C.items():
        d = dict(c)
        d["launch_us_under_pmc_median"] = us[k]
        d["samples_per_launch"] = V
        if "TCC_HIT_sum" in c and c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0) > 0:
            d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        if "TCP_TCC_READ_REQ_sum" in c:
            d["l2_read_requests_per_sample"] = c["TCP_TCC_READ_REQ_sum"] / V
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
            d["l1_accesses_per_sample"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / V
        if "FETCH_SIZE" in c:
            d["fabric_fetch_bytes_per_launch_x2_corrected"] = c["FETCH_SIZE"] * 1024.0 * 2.0
        enc[k] = d
    # the same kernel on the canonical samples of a real frame (ray-major): the LAST four k_encode_xcd launches of the pass
    Cc, usc, _ = per_kernel_tail(os.path.join(root, "pmc_encc_*", "**", "*counter_collection.csv"), "k_encode_xcd", 4)
    if Cc:
        d = dict(Cc)
        d["launch_us_under_pmc_median"] = usc
        nsamp = None
        try:
            for f in glob.glob(os.path.join(root, "pmc_encc_*.log")):
                for line in open(f):
                    if line.startswith("frame-coherent samples:"):
                        nsamp = int(line.split(":")[1])
        except Exception:
            pass
        d["samples_per_launch"] = nsamp
        if nsamp and "TCP_TCC_READ_REQ_sum" in d:
            d["l2_read_requests_per_sample"] = d["TCP_TCC_READ_REQ_sum"] / nsamp
        if nsamp and "TCP_TOTAL_CACHE_ACCESSES_sum" in d:
            d["l1_accesses_per_sample"] = d["TCP_TOTAL_CACHE_ACCESSES_sum"] / nsamp
        if "TCC_HIT_sum" in d and d["TCC_HIT_sum"] + d.get("TCC_MISS_sum", 0) > 0:
            d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
        enc["k_encode_xcd<16>/frame_coherent"] = d
    enc["_note"] = ("one pass per counter set over tools/pmc_encode.py (2^20 uniformly random points in the field bbox, 16-level table). "
                    "Algorithmic bytes: 512 B/sample.  L2 request-rate ceiling: 8 XCD x 16 channels x 2.1 GHz = 269 G requests/s "
                    "(profiles/r02_ubench_l2gather.txt).  FETCH_SIZE x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md)")
    enc.update(meta)
    json.dump(enc, open(os.path.join(out_dir, ROUND + "_pmc_encode.json"), "w"), indent=1)
    # ---- hash-grid backward atomics
    C, us, n = per_kernel(os.path.join(root, "pmc3_*", "**", "*counter_collection.csv"), ("k_hashgrid_bwd",))
    hg = {}
    for k, c in C.items():
        if "TCC_ATOMIC_sum" not in c:
            continue
        rate = c["TCC_ATOMIC_sum"] / (us[k] * 1e-6) / 1e9
        hg[k] = {"tcc_atomic_requests_per_launch": c["TCC_ATOMIC_sum"], "launch_us_under_pmc_median": us[k], "launches": n[k],
                 "atomic_requests_per_s_G_median": rate, "ceiling_G_per_s": 21.1, "frac_of_ceiling_median": rate / 21.1,
                 "tcc_ea0_atomic_per_launch": c.get("TCC_EA0_ATOMIC_sum")}
    hg["_note"] = ("TCC_ATOMIC_sum / TCC_EA0_ATOMIC_sum over an eager `bench.py --train-only` run (PatchSampler workload); ceiling = "
                   "tools/ubench/atomics.hip (21.1 G requests/s, profiles/r02_ubench_atomics.txt)")
    hg.update(meta)
    json.dump(hg, open(os.path.join(out_dir, ROUND + "_pmc_hgbwd.json"), "w"), indent=1)
    for n_ in (ROUND + "_pmc_encode.json", ROUND + "_pmc_hgbwd.json"):
        print("==", n_)
        print(open(os.path.join(out_dir, n_)).read()[:2500])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
