"""Builds the REFERENCE's own in-tree CUDA extensions for gfx950 into oracle/_ref/
(test infrastructure; oracle/_ref/ is git-ignored and never shipped).

The four extensions the reference JIT-builds at import time
(instant_avatar/deformers/fast_snarf/deformer_torch.py:10-19,
 instant_avatar/renderers/raymarcher_acc.py:13-16) are compiled UNMODIFIED from
/root/reference through torch.utils.cpp_extension (its hipify pass + hipcc;
works without a GPU).  torch's hipify writes next to the sources, and
/root/reference must not be written to, so the sources are staged in a scratch
directory under /tmp first (never inside this repo).  One portability patch is
applied to the scratch copy of raymarcher.cu only: the deprecated
`X.type()` dispatch argument -> `X.scalar_type()` (raymarcher.cu:93,178,248),
which current ATen no longer accepts.  No arithmetic is touched.

The resulting .so files let tests/test_ref_pin.py run the reference kernels on the
MI355X box and pin oracle/ia_oracle.c against them (and freeze golden vectors).
They are the reference's code: nothing under instantavatar_amd/ may load them.
"""
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF = "/root/reference/instant_avatar"
EXTS = {
    "ref_fuse_cuda": ["deformers/fast_snarf/cuda/fuse_kernel/fuse_cuda.cpp",
                      "deformers/fast_snarf/cuda/fuse_kernel/fuse_cuda_kernel_fast.cu"],
    "ref_filter": ["deformers/fast_snarf/cuda/filter/filter.cpp", "deformers/fast_snarf/cuda/filter/filter.cu"],
    "ref_precompute": ["deformers/fast_snarf/cuda/precompute/precompute.cpp",
                       "deformers/fast_snarf/cuda/precompute/precompute.cu"],
    "ref_raymarch": ["renderers/cuda/raymarcher.cpp", "renderers/cuda/raymarcher.cu"],
    # pytorch3d's CPU KNN (plain C++, no hipify): the reference source + our binding stub oracle/ref_knn_bind.cpp
    "ref_knn": ["../third_parties/pytorch3d/cuda/knn_cpu.cpp", "@oracle/ref_knn_bind.cpp"],
}


def build(force=False):
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree not present (only the prebuilt oracle/_ref/*.so travel to the GPU box)")
    os.makedirs(OUT, exist_ok=True)
    os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
    os.environ.setdefault("MAX_JOBS", "8")
    from torch.utils.cpp_extension import load
    built = []
    for name, srcs in EXTS.items():
        dst = os.path.join(OUT, name + ".so")
        if os.path.exists(dst) and not force:
            built.append(dst)
            continue
        stage = tempfile.mkdtemp(prefix="ia_ref_%s_" % name, dir="/tmp")
        staged = []
        for s in srcs:
            t = os.path.join(stage, os.path.basename(s))
            shutil.copy(os.path.join(HERE, os.path.basename(s)) if s.startswith("@oracle/") else os.path.join(REF, s), t)
            if t.endswith("raymarcher.cu"):
                txt = open(t).read()
                for v in ("rays_o", "sigma_vals"):
                    txt = txt.replace("AT_DISPATCH_FLOATING_TYPES_AND_HALF(%s.type()," % v,
                                      "AT_DISPATCH_FLOATING_TYPES_AND_HALF(%s.scalar_type()," % v)
                open(t, "w").write(txt)
            staged.append(t)
        bdir = os.path.join(stage, "build")
        os.makedirs(bdir)
        load(name=name, sources=staged, build_directory=bdir, verbose=False, is_python_module=False)
        so = os.path.join(bdir, name + ".so")
        shutil.copy(so, dst)
        shutil.rmtree(stage, ignore_errors=True)
        built.append(dst)
    return built


def load_ext(name):
    """Import a built reference extension as a python module (GPU box or here)."""
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded first)
    path = os.path.join(OUT, name + ".so")
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    for p in build(force="--force" in sys.argv):
        print(p)
