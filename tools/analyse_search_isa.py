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


def device_isa(flags=()):
    """the gfx950 assembly of ia_search.hip as a list of lines (the shipped per-unit flags applied)"""
    sys.path.insert(0, ROOT)
    from instantavatar_amd import build
    src = os.path.join(ROOT, "instantavatar_amd", "csrc", "ia_search.hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = [build.HIPCC] + [f for f in build.FLAGS if f != "-fPIC"] + list(build.TU_FLAGS.get("ia_search.hip", [])) + list(flags) + [
            "-x", "hip", "--cuda-device-only", "-S", src, "-o", out]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        return open(out).read().splitlines()


def dpp_hazards(text):
    """a DPP read needs two wait states after a VALU write of its source register, and inline asm is opaque to the
    compiler's hazard recogniser.  Walks every instruction stream of the file: for each `*_dpp` instruction (or one carrying a
    quad_perm / row_* modifier) the VGPRs it READS through DPP (src0) must not have been written by a VALU instruction in the
    two preceding wait states (a VALU instruction = 1, `s_nop N` = N + 1; labels and branches end the look-back conservatively:
    nothing is assumed across them, a hazard INSIDE a straight-line run is what inline asm can create).
    Returns a list of (line number, instruction, offending earlier instruction)."""
    bad = []
    hist = []          # [(wait states this instruction provides, set of vgprs it writes, text)] of the current straight-line run
    reg = re.compile(r"v(\d+)|v\[(\d+):(\d+)\]")

    def regs(tok):
        m = reg.fullmatch(tok.strip().rstrip(","))
        if not m:
            return set()
        if m.group(1) is not None:
            return {int(m.group(1))}
        return set(range(int(m.group(2)), int(m.group(3)) + 1))
    for ln, l in enumerate(text, 1):
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        if t.endswith(":") or t.startswith("s_cbranch") or t.startswith("s_branch") or t.startswith("s_setpc") or t.startswith("s_endpgm"):
            hist = []
            continue
        m = re.match(r"^([a-z][a-z0-9_]+)\s*(.*)$", t)
        if not m:
            continue
        op, rest = m.group(1), m.group(2).split(";")[0]
        ops = [o for o in re.split(r",\s*|\s+", rest) if o]
        is_dpp = op.endswith("_dpp") or " quad_perm:" in l or " row_shr:" in l or " row_shl:" in l or " row_bcast" in l or " row_mirror" in l
        if is_dpp and len(ops) >= 2:
            src = regs(ops[1])                                     # vdst, src0 (the DPP-permuted operand), ...
            need = 2
            for ws, wr, txt in reversed(hist):
                if need <= 0:
                    break
                if wr & src:
                    bad.append((ln, t, txt))
                    break
                need -= ws
        if op == "s_nop":
            hist.append((int(ops[0], 0) + 1 if ops else 1, set(), t))
        elif op.startswith("v_"):
            hist.append((1, regs(ops[0]) if ops else set(), t))
        elif op.startswith("s_") or op.startswith("ds_") or op.startswith("global_") or op.startswith("buffer_") or op.startswith("scratch_") or op.startswith("flat_"):
            hist.append((1, set(), t))                              # any other instruction is at least one wait state
        hist = hist[-4:]
    return bad


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


if __name__ == "__main__" and "--dpp-hazards" in sys.argv:
    bad = dpp_hazards(device_isa([a for a in sys.argv[1:] if a != "--dpp-hazards"]))
    print("DPP read-after-VALU-write hazards in ia_search.hip's device code: %d" % len(bad))
    for b in bad[:20]:
        print("  line %d: %s   <-   %s" % b)
    sys.exit(1 if bad else 0)
if __name__ == "__main__":
    main(sys.argv[1:])
