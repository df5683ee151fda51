"""Builds libinstantavatar_hip.so for gfx950 with hipcc (in-tree, next to the
sources, so the .so travels to the GPU box with the repo snapshot)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libinstantavatar_hip.so")
SOURCES = ["ia_error.cpp", "ia_snarf.hip", "ia_field.hip", "ia_render.hip", "ia_prof.hip", "ia_voxelise.hip", "ia_loss.hip", "ia_smpl_nn.hip", "ia_data.hip", "ia_mesh.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + [os.path.join(CSRC, "ia_common.h"), os.path.join(HERE, "..", "include", "instantavatar_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    objs = []
    for src in sources():
        obj = os.path.join(CSRC, os.path.basename(src) + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(
                os.path.getmtime(src), os.path.getmtime(os.path.join(CSRC, "ia_common.h")),
                os.path.getmtime(os.path.join(HERE, "..", "include", "instantavatar_hip.h"))):
            # -ffp-contract=off: fused multiply-adds appear only where the sources spell them
            # (IA_DOT3 / __builtin_fmaf), the same sequence the CPU checker uses
            cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "hip", "-c", src,
                   "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
