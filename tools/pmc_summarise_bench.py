"""Summarises the rocprofv3 --pmc passes of tools/pmc_all.sh (gpurun_out/pmc2_*/r_counter_collection.csv):
per kernel family and launch -- counters averaged over the launches of a family, durations from the dispatch
timestamps of the same pass.  Writes (to <out_dir>, copied to profiles/ by hand):

  <round>_pmc_traffic.json   HBM bytes per launch (FETCH_SIZE x 2 on gfx950 as MI355X_MICROARCH.md prescribes, + WRITE_SIZE)
  <round>_pmc_search.json    k_search: L1 / L2 hit rates, requests, stall and issue counters, probe vs render launches
  <round>_pmc_mfma.json      MFMA busy cycles of k_field / k_field_bwd against the SIMD cycles of the launch
  <round>_pmc_encode.json    (only with pmc_encode passes) kept from tools/pmc_encode.py

    python tools/pmc_summarise_bench.py <gpurun_out dir> <out dir>
"""
import collections
import csv
import glob
import json
import os
import sys

FAMILIES = ("k_search", "k_encode_xcd", "k_field_bwd_reduce", "k_field_bwd", "k_field", "k_hashgrid_bwd", "k_march_compact",
            "k_composite_compact", "k_precompute", "k_occ_components_lds", "k_occ_union")
N_SIMD = 256 * 4


def family(name):
    base = name.split("(")[0]
    for f in FAMILIES:
        if f in base:
            return f
    return None


def load(root):
    """{(family, variant)}: {counter: [sum, launches]}, plus durations.  variant for k_search: probe / render by grid size."""
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, set()]))
    dur = collections.defaultdict(lambda: [0.0, set()])
    for f in sorted(glob.glob(os.path.join(root, "pmc2_*", "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            fam = family(r["Kernel_Name"])
            if fam is None:
                continue
            var = fam
            if fam == "k_search":
                var = "k_search/probe" if int(r["Grid_Size"]) > 1500000 else "k_search/render"
            key = (f, r["Dispatch_Id"])
            for v in {var, fam}:
                a = acc[v][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1].add(key)
                d = dur[(v, r["Counter_Name"])]
                if key not in d[1]:
                    d[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                    d[1].add(key)
    out = {}
    for v, cs in acc.items():
        out[v] = {}
        for c, (tot, keys) in cs.items():
            out[v][c] = {"per_launch": tot / len(keys), "launches": len(keys), "avg_ns": dur[(v, c)][0] / len(keys)}
    return out


def main(root, out_dir, prefix="r02", meta=None):
    """meta: dict merged into every JSON written (round 3: the sha256 of the library the counters were collected on)"""
    S = load(root)
    g = lambda fam, c, k="per_launch": S.get(fam, {}).get(c, {}).get(k)
    # ---- traffic
    traffic = {}
    for fam in ("k_search", "k_field", "k_encode_xcd", "k_hashgrid_bwd", "k_field_bwd", "k_precompute", "k_march_compact"):
        fs, wsz = g(fam, "FETCH_SIZE"), g(fam, "WRITE_SIZE")
        if fs is None:
            continue
        fb, wb = fs * 1024.0 * 2.0, (wsz or 0.0) * 1024.0
        traffic[fam] = {"launches": g(fam, "FETCH_SIZE", "launches"), "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                        "hbm_bytes_per_launch": fb + wb, "avg_launch_us_in_pass": g(fam, "FETCH_SIZE", "avg_ns") / 1e3}
    if "k_field" in traffic and "k_encode_xcd" in traffic:  # the field STAGE as bench.py brackets it = encode + MLP kernels
        e, f = traffic["k_encode_xcd"], traffic["k_field"]
        traffic["k_field_stage"] = {"hbm_bytes_per_launch": e["hbm_bytes_per_launch"] * e["launches"] / max(f["launches"], 1) + f["hbm_bytes_per_launch"]}
    traffic["_note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over `bench.py --steps 3 --warmup 2 --no-graph "
                        "--train-steps 20`; KiB counters x 1024; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, "
                        "MI355X_MICROARCH.md section HBM); WRITE_SIZE uncalibrated")
    # ---- search
    search = {}
    for v in ("k_search", "k_search/probe", "k_search/render"):
        if v not in S:
            continue
        d = {}
        acc_, l2req = g(v, "TCP_TOTAL_CACHE_ACCESSES_sum"), g(v, "TCP_TCC_READ_REQ_sum")
        if acc_ and l2req is not None:
            d.update(tcp_accesses_per_launch=acc_, tcp_to_l2_read_requests_per_launch=l2req, l1_hit_rate=1.0 - l2req / acc_)
        h, m = g(v, "TCC_HIT_sum"), g(v, "TCC_MISS_sum")
        if h is not None and m is not None and h + m > 0:
            d.update(l2_hit_rate=h / (h + m), l2_hits_per_launch=h, l2_misses_per_launch=m)
        for c in ("TCP_PENDING_STALL_CYCLES_sum", "TCP_TA_TCP_STATE_READ_sum", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                  "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "TCC_REQ_sum",
                  "TCC_EA0_RDREQ_sum", "GRBM_GUI_ACTIVE"):
            if g(v, c) is not None:
                d[c] = g(v, c)
        wc = g(v, "SQ_WAVE_CYCLES")
        if wc and g(v, "SQ_WAIT_ANY") is not None:
            d["wave_cycle_split"] = {"waiting(s_waitcnt/barrier)": g(v, "SQ_WAIT_ANY") / wc, "issue_stalled": (g(v, "SQ_WAIT_INST_ANY") or 0) / wc,
                                     "issuing": (g(v, "SQ_ACTIVE_INST_ANY") or 0) / wc}
        ns = g(v, "TCP_TOTAL_CACHE_ACCESSES_sum", "avg_ns") or g(v, "FETCH_SIZE", "avg_ns")
        if ns:
            d["avg_launch_us_in_pass"] = ns / 1e3
            if acc_:
                d["tcp_accesses_per_clk_per_cu_at_2.4GHz"] = acc_ / (ns * 2.4 * 256)
        d["launches"] = max((x.get("launches", 0) for x in S[v].values()), default=0)
        search[v] = d
    for k in ("k_encode_xcd", "k_march_compact"):
        if k in S:
            acc_, l2req = g(k, "TCP_TOTAL_CACHE_ACCESSES_sum"), g(k, "TCP_TCC_READ_REQ_sum")
            h, m = g(k, "TCC_HIT_sum"), g(k, "TCC_MISS_sum")
            search[k] = {"l1_hit_rate": (1.0 - l2req / acc_) if acc_ else None, "l2_hit_rate": (h / (h + m)) if h is not None and m is not None and h + m > 0 else None}
    search["_note"] = ("counters are sums over the chip per launch, averaged over the launches; registers / occupancy of the kernel: "
                       "ia_kernel_info(\"k_search\") of the same library, reported live by bench.py")
    # ---- MFMA
    mfma = {}
    for fam in ("k_field", "k_field_bwd"):
        b = g(fam, "SQ_VALU_MFMA_BUSY_CYCLES")
        if b is None:
            continue
        ns = g(fam, "SQ_VALU_MFMA_BUSY_CYCLES", "avg_ns")
        clk_ghz = 2.4  # nominal maximum: the utilisation below is a LOWER bound when the chip clocks down under load
        mfma[fam] = {"mfma_busy_cycles_per_launch": b, "avg_launch_us_in_pass": ns / 1e3, "launches": g(fam, "SQ_VALU_MFMA_BUSY_CYCLES", "launches"),
                     "sq_busy_cycles_per_launch": g(fam, "SQ_BUSY_CYCLES"), "sq_wave_cycles_per_launch": g(fam, "SQ_WAVE_CYCLES"),
                     "clock_ghz_assumed": clk_ghz,
                     "mfma_utilisation": b / (ns * clk_ghz * N_SIMD) if ns else None}
    mfma["_note"] = ("SQ_VALU_MFMA_BUSY_CYCLES (32 per v_mfma_f32_32x32x16_f16, summed over the chip) / (launch duration x 2.4 GHz x 1024 SIMDs); the MLPs are 18 432 FLOP per "
                     "sample (22 v_mfma_f32_32x32x16_f16 per 32 samples forward): the matrix cores are used for the dense tiny-MLP GEMMs "
                     "only and are nowhere near a bound -- reported, not optimised (SURVEY 8d)")
    for name, obj in ((prefix + "_pmc_traffic.json", traffic), (prefix + "_pmc_search.json", search), (prefix + "_pmc_mfma.json", mfma)):
        obj.update(meta or {})
        json.dump(obj, open(os.path.join(out_dir, name), "w"), indent=1)
        print("==", name)
        print(json.dumps(obj, indent=1)[:3000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
