"""Generates tests/golden/novel_view_golden.npz: the REFERENCE's novel_view.py executing on the CPU -- its AnimateDataset
(camera :29-44, the fixed pose :46-49, the per-frame turn about y :80-85, near / far :88-89) at downscale 16 (67 x 67 rays),
12 frames.  hydra / lightning / tqdm / imageio are empty stand-ins; cv2 is not installed either, and the data set calls
`cv2.Rodrigues` (matrix <-> rotation vector): the stand-in routes it to scipy.spatial.transform.Rotation, an implementation
independent of the package's own Rodrigues code.  (A rotation by exactly pi has two equivalent vectors, +-pi k: the test
compares rotation MATRICES.)
Run from the repo root:  python tests/golden/make_novel_view_golden.py"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "novel_view_golden.npz")
REF = "/root/reference"
N_FRAMES, DOWNSCALE = 12, 16


def rodrigues(x):
    from scipy.spatial.transform import Rotation
    x = np.asarray(x, np.float64)
    if x.shape == (3, 3):
        return Rotation.from_matrix(x).as_rotvec().reshape(3, 1), None
    return Rotation.from_rotvec(x.reshape(3)).as_matrix(), None


def main():
    for name in ("pytorch_lightning", "tqdm", "imageio"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["tqdm"].tqdm = lambda x, *a, **k: x
    cv2 = types.ModuleType("cv2")
    cv2.Rodrigues = rodrigues
    sys.modules["cv2"] = cv2
    hydra = types.ModuleType("hydra")
    hydra.main = lambda **kw: (lambda f: f)
    sys.modules["hydra"] = hydra
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import novel_view as ref
    betas = (np.arange(10, dtype=np.float32) - 4.5) * 0.1
    ds = ref.AnimateDataset(N_FRAMES, betas=betas[None], downscale=DOWNSCALE)
    out = dict(H=np.int32(ds.H), W=np.int32(ds.W), n=np.int32(len(ds)), betas=betas)
    for f in range(N_FRAMES):
        d = ds[f]
        for k, v in d.items():
            if f == 0 or k == "global_orient":
                out["f%d_%s" % (f, k)] = np.asarray(v)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes; H, W, frames:", ds.H, ds.W, len(ds))


if __name__ == "__main__":
    main()
