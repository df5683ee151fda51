"""Generates tests/golden/loss_golden.npz: the REFERENCE's instant_avatar/utils/loss.py executing on the CPU -- NeRFLoss
(:53-77) on a flat 512-ray batch and NGPLoss (:8-50, depth regulariser on, LPIPS weight 0) on a patch batch
[1, 2, 16, 16, *]: every reported value and the gradients w.r.t. the four predicted tensors.  torchvision (needed by the
LPIPS module NGPLoss constructs) is an architecture-only stand-in, see make_lpips_golden.py.
Run from the repo root:  python tests/golden/make_loss_golden.py"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
OUT = os.path.join(HERE, "loss_golden.npz")
REF = "/root/reference"


class Opt(dict):
    __getattr__ = dict.__getitem__


def inputs(seed, shape):
    g = torch.Generator().manual_seed(seed)
    n = lambda *s: torch.rand(*s, generator=g)
    pred = {"rgb_coarse": n(*shape, 3), "alpha_coarse": n(*shape), "depth_coarse": n(*shape) * 3 + 2, "weight_coarse": n(*shape, 24) * 0.2}
    tgt = {"rgb": n(*shape, 3), "alpha": (n(*shape) > 0.5).float()}
    return pred, tgt


def main():
    import make_lpips_golden as mk
    tv = types.ModuleType("torchvision")
    models = types.ModuleType("torchvision.models")
    models.VGG16_Weights = types.SimpleNamespace(DEFAULT=None)
    models.vgg16 = lambda weights=None: types.SimpleNamespace(features=mk.vgg16_features())
    tv.models = models
    sys.modules["torchvision"], sys.modules["torchvision.models"] = tv, models
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import third_parties.lpips as ref_pkg
    import third_parties.lpips.lpips as ref_mod
    if not hasattr(ref_mod, "normalize_tensor"):
        ref_mod.normalize_tensor = ref_pkg.normalize_tensor
    import instant_avatar.utils.loss as L
    out = {}
    for tag, cls, opt, shape in (("nerf", L.NeRFLoss, Opt(w_rgb=1.0, w_alpha=0.1, w_reg=0.1), (1, 512)),
                                 ("ngp", L.NGPLoss, Opt(w_rgb=1.0, w_alpha=0.1, w_reg=0.1, w_lpips=0.0, w_depth_reg=0.01), (1, 2, 16, 16))):
        pred, tgt = inputs(31 if tag == "nerf" else 32, shape)
        for v in pred.values():
            v.requires_grad_(True)
        losses = cls(opt)(pred, tgt)
        losses["loss"].backward()
        for k, v in losses.items():
            out["%s_%s" % (tag, k)] = np.float64(float(v))
        for k, v in pred.items():
            out["%s_in_%s" % (tag, k)] = v.detach().numpy()
            out["%s_grad_%s" % (tag, k)] = (v.grad if v.grad is not None else torch.zeros_like(v)).numpy()
        for k, v in tgt.items():
            out["%s_tgt_%s" % (tag, k)] = v.numpy()
        print(tag, {k: float(v) for k, v in losses.items()})
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
