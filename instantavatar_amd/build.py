"""Builds libinstantavatar_hip.so for gfx950 with hipcc (in-tree, next to the
sources, so the .so travels to the GPU box with the repo snapshot).

The library carries a MANIFEST of what it was built from, two 16-hex-digit hashes per translation unit:
  * source hash over (compiler flags, the two shared headers, local includes, the source file) -- `needs_build()`
    compares it with the checkout as it is now (not file times: a prebuilt library that is newer than an edited
    checkout used to win);
  * device-code hash = sha256 of the object's `.hip_fatbin` section (the gfx950 code objects) -- `bench.py` accepts a
    committed PMC summary under profiles/ only when the device code of the translation unit that holds the profiled
    kernel equals the running library's.  The evidence therefore survives a rebuild of the same sources on another box
    or path, a comment or a new declaration in the header, and an edit of ANOTHER translation unit (a summary of
    k_search stays valid while only ia_render.hip changes); it does not survive a change of the kernel itself.
"""
import concurrent.futures
import hashlib
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libinstantavatar_hip.so")
SOURCES = ["ia_error.cpp", "ia_snarf.hip", "ia_search.hip", "ia_field.hip", "ia_render.hip", "ia_prof.hip", "ia_voxelise.hip", "ia_loss.hip", "ia_smpl_nn.hip", "ia_data.hip", "ia_mesh.hip", "ia_optim.hip", "ia_smpl_lbs.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: fused multiply-adds appear only where the sources spell them
# (IA_DOT3 / __builtin_fmaf), the same sequence the CPU checker uses
# -cuid=<translation unit name> (cuid_flag): hipcc's default derives the `__hip_cuid_*` symbol from the source PATH;
# named after the file instead, the objects (host AND device code) of the same sources are byte-identical whatever
# directory they are built in, which is what lets a counter summary name the device code it was collected on
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]
# per-translation-unit flags.  ia_search.hip: the SLP vectoriser pairs scalar fp32 ops of the Broyden update into v_pk_* and
# pays with register moves (static solver-loop count 614 -> 606 VALU, 51 -> 29 moves; per-lane arithmetic identical, the
# parity tests are bit-exact with it): 193.4 -> 191.7 us on a frame's sample points (profiles/r04_ab_search_variants.txt)
TU_FLAGS = {"ia_search.hip": ["-fno-slp-vectorize"]}
OBJCOPY = os.environ.get("LLVM_OBJCOPY", "/opt/rocm/lib/llvm/bin/llvm-objcopy")
SHARED_HEADERS = [os.path.join(CSRC, "ia_common.h"), os.path.join(HERE, "..", "include", "instantavatar_hip.h")]
_MARK = b"IA_SOURCE_MANIFEST="


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def extra_flags(src=None):
    """experiment switches for A/B builds (tools/ab_*.sh): IA_EXTRA_HIPCC_FLAGS="[<file>:]<flags>" -- with a file prefix
    (e.g. "ia_snarf.hip:-DIA_X=1") only that translation unit gets them.  Part of the unit's hashes, so a variant never
    passes for the default build."""
    own = list(TU_FLAGS.get(os.path.basename(src), [])) if src is not None else []
    v = os.environ.get("IA_EXTRA_HIPCC_FLAGS", "").strip()
    if not v:
        return own
    head = v.split()[0]
    if ":" in head and head.split(":")[0].endswith((".hip", ".cpp")):
        only, rest = v.split(":", 1)
        return own + (rest.split() if src is not None and os.path.basename(src) == only else [])
    return own + v.split()


def cuid_flag(src):
    return "-cuid=" + re.sub(r"[^A-Za-z0-9]", "_", os.path.basename(src))


def includes_of(src):
    """csrc-local headers a translation unit includes (e.g. the kernels' .inc pieces), transitively"""
    seen, todo = [], [src]
    while todo:
        f = todo.pop()
        for m in re.finditer(r'^\s*#\s*include\s+"([^"]+)"', open(f).read(), re.M):
            p = os.path.normpath(os.path.join(os.path.dirname(f), m.group(1)))
            if os.path.exists(p) and p not in seen and os.path.normpath(p) not in [os.path.normpath(h) for h in SHARED_HEADERS]:
                seen.append(p)
                todo.append(p)
    return sorted(seen)


def tu_hash(src):
    h = hashlib.sha256()
    h.update(" ".join(FLAGS + extra_flags(src) + [cuid_flag(src)]).encode())
    for f in SHARED_HEADERS + includes_of(src) + [src]:
        h.update(b"\0" + os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def source_manifest():
    """{translation unit: hash} of the checkout"""
    return {os.path.basename(s): tu_hash(s) for s in sources()}


def _manifest_string(m):
    return ";".join("%s=%s" % kv for kv in sorted(m.items()))


def _raw_manifest(path=None):
    path = path or OUT
    if not os.path.exists(path):
        return {}
    blob = open(path, "rb").read()
    i = blob.find(_MARK)
    if i < 0:
        return {}
    j = blob.find(b"\0", i)
    s = blob[i + len(_MARK):j].decode("ascii", "replace")
    return dict(kv.split("=", 1) for kv in s.split(";") if "=" in kv)


def library_manifest(path=None):
    """{translation unit: source hash} the library at `path` was built from, read from its bytes (no dlopen: the file
    may be about to be replaced); {} when there is no library or it predates the manifest"""
    return {k: v.split(":")[0] for k, v in _raw_manifest(path).items()}


def device_manifest(path=None):
    """{translation unit: hash of its gfx950 code objects} of the library at `path`"""
    return {k: v.split(":")[1] for k, v in _raw_manifest(path).items() if ":" in v}


def device_code_hash(obj):
    """sha256 (16 hex digits) of the `.hip_fatbin` section of an object; "-" for host-only objects"""
    tmp = obj + ".fatbin.%d" % os.getpid()
    try:
        subprocess.check_call([OBJCOPY, "-O", "binary", "--only-section=.hip_fatbin", obj, tmp],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        blob = open(tmp, "rb").read()
    except Exception:
        return "?"
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return hashlib.sha256(blob).hexdigest()[:16] if blob else "-"


def have_compiler():
    return os.path.exists(HIPCC) or shutil.which(HIPCC) is not None


def needs_build():
    if not os.path.exists(OUT):
        return True
    return library_manifest() != source_manifest()


def ensure_current(verbose=False):
    """Entry points that are started on a copy of the tree (pytest on the GPU box, bench.py, __graft_entry__.smoke) call this first:
    a library that is missing or was built from other sources than this checkout is REBUILT when the compiler is there (one process
    at a time: an exclusive lock file next to the library; the others wait and find it current), and left alone otherwise --
    `_lib.lib()` then refuses it loudly.  Not a fallback: the product still only ever runs the HIP library of this checkout."""
    if os.environ.get("IA_EXTRA_HIPCC_FLAGS") or os.environ.get("IA_ALLOW_STALE_LIB") == "1":
        return False      # an A/B variant is being run on purpose (tools/ab_lib.sh)
    if not needs_build() or not have_compiler():
        return False
    import fcntl
    with open(OUT + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if needs_build():
                if verbose:
                    print("instantavatar_amd: library missing or stale, rebuilding it from this checkout", file=sys.stderr)
                build(force=False, verbose=False)
                return True
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return False


def build(force=False, verbose=False):
    """Compile what changed (everything with force=True) and link.  Without a compiler a library whose manifest equals
    the checkout is used as it is; a stale one is an error, never silently run."""
    if not force and not needs_build():
        return OUT
    if not have_compiler():
        if os.path.exists(OUT) and not needs_build():
            return OUT
        raise RuntimeError("libinstantavatar_hip.so is missing or was built from other sources than this checkout, and %s "
                           "is not available to rebuild it" % HIPCC)
    want = source_manifest()
    stamp_dir = os.path.join(CSRC, ".stamps")
    os.makedirs(stamp_dir, exist_ok=True)
    jobs, objs = [], []
    for src in sources():
        base = os.path.basename(src)
        obj = os.path.join(CSRC, base + ".o")
        stamp = os.path.join(stamp_dir, base + ".hash")
        objs.append(obj)
        if base == "ia_error.cpp":
            continue    # carries the manifest: compiled last, always
        have = open(stamp).read().strip() if os.path.exists(stamp) else ""
        if force or not os.path.exists(obj) or have != want[base]:
            jobs.append((src, obj, stamp, want[base]))

    def compile_one(job):
        src, obj, stamp, h = job
        cmd = [HIPCC] + FLAGS + extra_flags(src) + [cuid_flag(src), "-x", "hip", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        open(stamp, "w").write(h)

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, max(1, os.cpu_count() or 1))) as ex:
        list(ex.map(compile_one, jobs))
    err_src = os.path.join(CSRC, "ia_error.cpp")
    full = {k: "%s:%s" % (v, device_code_hash(os.path.join(CSRC, k + ".o")) if k != "ia_error.cpp" else "-") for k, v in want.items()}
    cmd = [HIPCC] + FLAGS + extra_flags(err_src) + [cuid_flag(err_src), '-DIA_SOURCE_MANIFEST="%s"' % _manifest_string(full), "-x", "hip", "-c", err_src,
                                              "-o", os.path.join(CSRC, "ia_error.cpp.o")]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    tmp = OUT + ".tmp.%d" % os.getpid()
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, OUT)      # atomic: a process that has the old library mapped keeps its (unlinked) file
    got = library_manifest()
    assert got == want, "the library does not carry the manifest it was built with: %r vs %r" % (got, want)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(_manifest_string(_raw_manifest()))
