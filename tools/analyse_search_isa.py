"""Dev tool (no GPU needed): the instruction mix of k_search's solver loop, from the ISA hipcc generates for gfx950.

    python tools/analyse_search_isa.py                 # the shipped configuration
    python tools/analyse_search_isa.py -DIA_QUAD_ASM_DPP_ADD=1 -DIA_PLAN_FACTOR_ZERO=1

Compiles instantavatar_amd/csrc/ia_search.hip to assembly (device only), cuts out the lane state machine of k_search<1> (the
loop that holds the global_load_dwordx4 of the trilinear fetch) and counts its instructions by class.  The counts are STATIC
(every path of the loop body once: refill, first iteration, update, done); the dynamic count per wave-step is lower (the
refill and done paths are taken rarely) -- profiles/r03_pmc_issue/ has the executed totals."""
import collections
import re
import subprocess
import sys
import tempfile
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLASSES = [
    ("vector loads (16 B)", r"^global_load_dwordx4"),
    ("other global / scratch memory", r"^(global_|scratch_|buffer_)"),
    ("LDS", r"^ds_"),
    ("DPP moves / DPP-folded ops", r"_dpp$"),
    ("packed fp32 fma / mul / add", r"^v_pk_"),
    ("division helpers (div_scale, div_fmas, div_fixup, rcp, frexp)", r"^v_(div_|rcp_|frexp_)"),
    ("fp32 fma / mul / add / sub", r"^v_(fma|fmac|mul|add|sub|mac)_f32"),
    ("integer multiply (24-bit, full rate)", r"^v_(mul|mad)_[iu]32_[iu]24"),
    ("integer multiply (32-bit, quarter rate)", r"^v_mul_(lo|hi)_[iu]32"),
    ("integer add / shift / logic", r"^v_(add|sub|lshl|lshr|ashr|and|or|xor|not|bfe|add3|lshl_add|and_or|min|max|med3)_?[a-z0-9_]*[iub](16|32|64)"),
    ("conversions / floor", r"^v_(cvt_|floor_|trunc_|rndne_)"),
    ("compares", r"^v_cmp"),
    ("selects", r"^v_cndmask"),
    ("register moves", r"^v_(mov|pk_mov|accvgpr|readlane|readfirstlane|writelane)"),
    ("cross-lane (bpermute / permlane / swizzle)", r"^(ds_bpermute|ds_permute|ds_swizzle|v_permlane)"),
    ("other VALU", r"^v_"),
    ("s_waitcnt / s_nop", r"^s_(waitcnt|nop)"),
    ("scalar branches", r"^s_(cbranch|branch)"),
    ("scalar mask / ALU", r"^s_"),
]


def main(flags):
    src = os.path.join(ROOT, "instantavatar_amd", "csrc", "ia_search.hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", *flags, "-x", "hip",
               "--cuda-device-only", "-S", src, "-o", out]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    start = next(i for i, l in enumerate(text) if l.startswith("_Z8k_searchILi1E"))
    end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
    body = text[start:end]
    res = {k: next((l for l in text[end:end + 400] if k in l), "") for k in ("; NumVgprs:", "; LDSByteSize:", "; Occupancy:", "; ScratchSize:")}
    # the solver loop = the innermost-level loop (as labelled by the compiler) that contains the 16-byte loads
    heads = [i for i, l in enumerate(body) if "Loop Header" in l]
    loads = [i for i, l in enumerate(body) if "global_load_dwordx4" in l]
    h = max(x for x in heads if x < loads[0])
    nxt = min([x for x in heads if x > loads[-1]] + [len(body)])
    loop = body[h:nxt]
    counts = collections.OrderedDict((name, 0) for name, _ in CLASSES)
    total = 0
    for l in loop:
        m = re.match(r"^\s+([a-z][a-z0-9_]+)", l)
        if not m:
            continue
        op = m.group(1)
        if op.endswith("_e32") or op.endswith("_e64"):
            op = op[:-4]
        if " quad_perm:" in l or " row_shr:" in l or " row_bcast" in l:
            op = op + "_dpp" if not op.endswith("_dpp") else op
        for name, pat in CLASSES:
            if re.search(pat, op):
                counts[name] += 1
                total += 1
                break
    print("k_search<1>, flags %s:  %s %s %s %s" % (" ".join(flags) or "(default)", *[res[k].strip("; ").strip() for k in res]))
    print("solver loop, static instruction counts (%d lines of ISA, %d instructions):" % (len(loop), total))
    valu = sum(v for k, v in counts.items() if not (k.startswith("s_") or k.startswith("scalar") or k in ("LDS", "vector loads (16 B)", "other global / scratch memory", "cross-lane (bpermute / permlane / swizzle)")))
    for k, v in counts.items():
        if v:
            print("  %-68s %4d" % (k, v))
    print("  %-68s %4d" % ("== VALU total", valu))


if __name__ == "__main__":
    main(sys.argv[1:])
