"""Generates tests/golden/animate_golden.npz: the REFERENCE's animate.py executing on the CPU -- AnimateDataset (camera of
animate.py:27-44, pose track handling :46-54, __getitem__ :59-80) on the pose track it ships
(data/animation/aist_demo.npz), at downscale 16 (67 x 67 rays).  hydra / lightning / cv2 / tqdm / imageio are empty
stand-ins (`@hydra.main` becomes the identity decorator); nothing of them is used by the dataset.
Run from the repo root:  python tests/golden/make_animate_golden.py"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "animate_golden.npz")
REF = "/root/reference"
FRAMES, DOWNSCALE = (0, 17, 319), 16


def main():
    for name in ("cv2", "pytorch_lightning", "tqdm", "imageio"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["tqdm"].tqdm = lambda x, *a, **k: x
    hydra = types.ModuleType("hydra")
    hydra.main = lambda **kw: (lambda f: f)
    sys.modules["hydra"] = hydra
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import animate as ref
    path = os.path.join(REF, "data/animation/aist_demo.npz")
    betas = (np.arange(10, dtype=np.float32) - 4.5) * 0.1
    ds = ref.AnimateDataset(path, betas=betas[None], downscale=DOWNSCALE)
    out = dict(H=np.int32(ds.H), W=np.int32(ds.W), n=np.int32(len(ds)), betas=betas, frames=np.array(FRAMES))
    track = np.load(path)
    out["track_poses"], out["track_trans"] = track["poses"][list(FRAMES)], track["trans"][list(FRAMES)]   # FRAMES[0] == 0: the origin row
    for f in FRAMES:
        d = ds[f]
        for k, v in d.items():
            out["f%d_%s" % (f, k)] = np.asarray(v)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes; H, W, frames:", ds.H, ds.W, len(ds))


if __name__ == "__main__":
    main()
