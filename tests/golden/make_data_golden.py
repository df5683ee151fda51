"""Generates tests/golden/data_golden.npz: the REFERENCE's data side executing on the CPU --
instant_avatar/datasets/peoplesnapshot.py (make_rays, PeopleSnapshotDataset.__getitem__, train split) and
instant_avatar/utils/sampler.py (EdgeSampler, PatchSampler incl. dilate) -- on a synthetic 96 x 80 frame.

Stand-ins (cv2 is not installed): cv2.erode / cv2.dilate with a k x k box = scipy.ndimage minimum / maximum filters with
OpenCV's anchor (k // 2) and "ignore pixels outside" border; cv2.imread returns the synthetic BGR image.  numpy's global
random functions are scripted: every call returns what the listed uniform draws map to under floor(u * count) (and, for
choice(replace=False), sequential draws from the remaining candidates) -- the mapping the device samplers and
oracle/data_oracle.py use, so that all three can be compared on identical draws.
Run from the repo root:  python tests/golden/make_data_golden.py"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
OUT = os.path.join(HERE, "data_golden.npz")
H, W, SEED = 96, 80, 5


def scene():
    rs = np.random.RandomState(SEED)
    yy, xx = np.mgrid[0:H, 0:W]
    mask = ((((yy - 50) / 30.0) ** 2 + ((xx - 38) / 17.0) ** 2) < 1).astype(np.float32)
    mask[20:30, 60:70] = 1           # a detached blob
    img = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
    K = np.array([[180.0, 0, W / 2], [0, 175.0, H / 2], [0, 0, 1]])
    c2w = np.eye(4)
    c2w[:3, :3] = [[0.98, -0.1, 0.17], [0.12, 0.99, -0.1], [-0.16, 0.12, 0.98]]
    c2w[:3, 3] = [0.1, -0.2, 0.3]
    smpl = dict(betas=rs.randn(1, 10).astype(np.float32), body_pose=rs.randn(4, 69).astype(np.float32) * 0.2,
                global_orient=rs.randn(4, 3).astype(np.float32) * 0.2, transl=(rs.randn(4, 3) * 0.3 + [0, 0.1, 4.0]).astype(np.float32))
    return mask, img, K, c2w, smpl


class Script:
    """np.random.{rand, randint, choice} fed from explicit uniform draws"""

    def __init__(self):
        self.q = []

    def feed(self, *arrays):
        self.q = [np.asarray(a, np.float32) for a in arrays]

    def rand(self, *shape):
        a = self.q.pop(0)
        if not shape:
            return float(a.reshape(-1)[0])
        return a.reshape(shape).astype(np.float64)

    def randint(self, lo, hi, size=None):
        u = self.q.pop(0).reshape(-1)
        n = int(size) if np.isscalar(size) else int(np.prod(size))
        assert lo == 0 and len(u) == n, (lo, len(u), n)
        return np.minimum(np.floor(u * np.float32(hi)).astype(np.int64), hi - 1)

    def choice(self, count, size, replace=True):
        assert not replace
        u = self.q.pop(0).reshape(-1)
        remaining, pick = list(range(count)), []
        for i in range(size):
            r = int(min(np.floor(u[i] * np.float32(len(remaining))), len(remaining) - 1))
            pick.append(remaining.pop(r))
        return np.asarray(pick)


def main():
    import scipy.ndimage as ndi
    import torch  # noqa: F401  (the reference modules import it)
    mask, img, K, c2w, smpl = scene()
    cv2 = types.ModuleType("cv2")

    def box(a, kernel, fn, fill):
        """k x k box morphology the way OpenCV runs it: anchor (k // 2, k // 2), pixels outside the image ignored; a 1-D
        array of length N is an N x 1 image (rows = N, one column) -- which is what EdgeSampler hands over, because it
        flattens the mask BEFORE the morphology (sampler.py:23-27): its band is computed along the flattened index."""
        k = kernel.shape[0]
        a = np.asarray(a, np.float32)
        img2d = a.reshape(-1, 1) if a.ndim == 1 else a
        return fn(img2d, footprint=np.ones((k, k), bool), mode="constant", cval=fill)
    cv2.erode = lambda a, kernel: box(a, kernel, ndi.minimum_filter, np.inf)
    cv2.dilate = lambda a, kernel: box(a, kernel, ndi.maximum_filter, -np.inf)
    cv2.imread = lambda path: img
    sys.modules["cv2"] = cv2
    hydra = types.ModuleType("hydra")
    sys.modules["hydra"] = hydra
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningDataModule = object
    sys.modules["pytorch_lightning"] = pl
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    import instant_avatar.utils.sampler as rs_mod
    import instant_avatar.datasets.peoplesnapshot as ds_mod
    script = Script()
    np.random.rand, np.random.randint, np.random.choice = script.rand, script.randint, script.choice
    out = {}
    ro, rd = ds_mod.make_rays(K, c2w, H, W)
    out["rays_o"], out["rays_d"] = ro, rd
    rs = np.random.RandomState(SEED + 1)
    # EdgeSampler, as written: it flattens the mask first, so cv2 sees an (H*W) x 1 image
    e_draws = rs.rand(512).astype(np.float32)
    es = rs_mod.EdgeSampler(num_sample=512, ratio_mask=0.6, ratio_edge=0.3, kernel_size=8)
    n_m, n_e = es.num_mask, es.num_edge
    script.feed(e_draws[:n_m], e_draws[n_m:n_m + n_e], e_draws[n_m + n_e:])
    e_out = es.sample(mask, img.astype(np.float32), ro, rd)
    out["edge_draws"] = e_draws
    for i, a in enumerate(e_out):
        out["edge_out%d" % i] = np.asarray(a)
    # PatchSampler: mask branch (with and without dilate) and uniform branch
    for tag, coin, dil in (("pm", 0.2, 0), ("pd", 0.2, 6), ("pu", 0.95, 0)):
        ps = rs_mod.PatchSampler(num_patch=4, patch_size=16, ratio_mask=0.9, dilate=dil)
        d = np.r_[coin, rs.rand(8)].astype(np.float32)
        if coin < 0.9:
            script.feed(d[:1], d[1:5])
        else:
            script.feed(d[:1], d[1:5], d[5:9])
        p_out = ps.sample(mask, img.astype(np.float32), ro, rd)
        out[tag + "_draws"] = d
        for i, a in enumerate(p_out):
            out["%s_out%d" % (tag, i)] = np.asarray(a)
    # __getitem__ (train) with the patch sampler
    ds = ds_mod.PeopleSnapshotDataset.__new__(ds_mod.PeopleSnapshotDataset)
    mpath = "/tmp/_ia_mask.npy"
    np.save(mpath, mask)
    ds.img_lists, ds.msk_lists, ds.downscale, ds.split = ["img0"], [mpath], 1, "train"
    ds.rays_o, ds.rays_d, ds.smpl_params, ds.near, ds.far = ro, rd, smpl, None, None
    ds.sampler = rs_mod.PatchSampler(num_patch=4, patch_size=16, ratio_mask=0.9, dilate=0)
    bg = rs.rand(H, W, 3).astype(np.float32)
    d = np.r_[0.1, rs.rand(8)].astype(np.float32)
    script.feed(bg, d[:1], d[1:5])
    datum = ds.__getitem__(0)
    out["gi_draws"], out["gi_bg"] = d, bg
    for k, v in datum.items():
        out["gi_" + k] = np.asarray(v)
    # __getitem__ of the "val" split: the whole frame, white background (peoplesnapshot.py:112-125)
    ds.split = "val"
    script.feed()
    datum = ds.__getitem__(2 if len(ds.img_lists) > 2 else 0)
    for k, v in datum.items():
        out["ge_" + k] = np.asarray(v)
    np.savez_compressed(OUT, H=np.int32(H), W=np.int32(W), seed=np.int32(SEED), **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", {k: np.asarray(v).shape for k, v in datum.items()})


if __name__ == "__main__":
    main()
