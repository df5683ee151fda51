"""Generates tests/golden/lpips_alex_golden.npz by running the REFERENCE's LPIPS module
(/root/reference/third_parties/lpips, net="alex", version 0.1, its own pretrained lin layers: weights/v0.1/alex.pth) on
seeded inputs -- the network behind eval.py's metric (eval.py:18: torchmetrics' LearnedPerceptualImagePatchSimilarity
(net_type="alex") wraps this same v0.1 module; torchmetrics itself is not installed here).

As for the VGG golden (make_lpips_golden.py): torchvision is not installed and its pretrained AlexNet weights could not be
downloaded, so the reference's trunk wrapper (pretrained_networks.alexnet) is given a stand-in `torchvision.models.alexnet`
with the standard feature stack and DETERMINISTIC closed-formula weights, which the test re-creates.  Two calls are
recorded: normalize=True (inputs in [0, 1] rescaled to [-1, 1]) and normalize=False on the SAME [0, 1] inputs -- what
eval.py effectively computes, because torchmetrics' default is normalize=False and eval.py hands it [0, 1] images.

Run from the repo root in the build container:  python tests/golden/make_lpips_alex_golden.py
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_lpips_golden import formula_weights  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(HERE, "lpips_alex_golden.npz")


def alexnet_features():
    return nn.Sequential(nn.Conv2d(3, 64, kernel_size=11, stride=4, padding=2), nn.ReLU(inplace=True), nn.MaxPool2d(kernel_size=3, stride=2),
                         nn.Conv2d(64, 192, kernel_size=5, padding=2), nn.ReLU(inplace=True), nn.MaxPool2d(kernel_size=3, stride=2),
                         nn.Conv2d(192, 384, kernel_size=3, padding=1), nn.ReLU(inplace=True),
                         nn.Conv2d(384, 256, kernel_size=3, padding=1), nn.ReLU(inplace=True),
                         nn.Conv2d(256, 256, kernel_size=3, padding=1), nn.ReLU(inplace=True), nn.MaxPool2d(kernel_size=3, stride=2))


def main():
    tv = types.ModuleType("torchvision")
    models = types.ModuleType("torchvision.models")

    class _W:
        DEFAULT = None
    models.AlexNet_Weights = _W
    models.alexnet = lambda weights=None: types.SimpleNamespace(features=alexnet_features())
    tv.models = models
    sys.modules["torchvision"], sys.modules["torchvision.models"] = tv, models
    sys.path.insert(0, REF)
    import third_parties.lpips as ref_pkg
    from third_parties.lpips import LPIPS
    import third_parties.lpips.lpips as ref_mod
    if not hasattr(ref_mod, "normalize_tensor"):          # circular star-import of the vendored package (see make_lpips_golden.py)
        ref_mod.normalize_tensor = ref_pkg.normalize_tensor
    m = LPIPS(net="alex", pretrained=True, pnet_rand=True, verbose=False)
    with torch.no_grad():
        for salt, (name, p) in enumerate(sorted(m.net.named_parameters())):
            p.copy_(formula_weights(tuple(p.shape), salt))
    g = torch.Generator().manual_seed(21)
    x = torch.rand((2, 3, 72, 64), generator=g)
    y = (x + 0.15 * torch.randn((2, 3, 72, 64), generator=g)).clamp(0, 1)
    with torch.no_grad():
        val, per = m(x, y, retPerLayer=True, normalize=True)
        val_raw, per_raw = m(x, y, retPerLayer=True, normalize=False)
    lin = {k: v.numpy() for k, v in m.state_dict().items() if k.startswith("lin")}
    np.savez_compressed(OUT, x=x.numpy(), y=y.numpy(), val=val.numpy(), per=np.stack([r.numpy() for r in per]),
                        val_raw=val_raw.numpy(), per_raw=np.stack([r.numpy() for r in per_raw]),
                        param_order=np.array(sorted(n for n, _ in m.net.named_parameters())), **{k.replace(".", "__"): v for k, v in lin.items()})
    print("wrote", OUT, os.path.getsize(OUT), "bytes; val", val.reshape(-1).tolist(), "raw", val_raw.reshape(-1).tolist())


if __name__ == "__main__":
    main()
