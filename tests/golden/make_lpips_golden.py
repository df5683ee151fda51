"""Generates tests/golden/lpips_golden.npz by running the REFERENCE's LPIPS module
(/root/reference/third_parties/lpips, net="vgg", version 0.1, its own pretrained lin layers) on seeded inputs.

torchvision is not installed in this image, and its pretrained VGG-16 weights could not be downloaded anyway: the
reference's trunk wrapper (pretrained_networks.vgg16) is given a stand-in `torchvision.models.vgg16` with the standard
configuration-D feature stack and DETERMINISTIC weights (a closed formula, below), which the test re-creates.  What the
golden pins is everything the reference's code does around the trunk: input scaling, the slice boundaries, channel
normalisation, squared differences, the pretrained 1x1 lin layers, spatial mean and layer sum.

Run from the repo root in the build container:  python tests/golden/make_lpips_golden.py
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lpips_golden.npz")


def formula_weights(shape, salt):
    """Deterministic pseudo-weights of He-like magnitude: sin of an affine function of the flat index."""
    n = int(np.prod(shape))
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
    i = np.arange(n, dtype=np.float64)
    v = np.sin(i * 12.9898 + salt * 78.233) * np.sqrt(2.0 / fan_in) * 1.7
    if len(shape) == 1:
        v = 0.05 * np.sin(i * 0.7 + salt)
    return torch.as_tensor(v.reshape(shape), dtype=torch.float32)


def vgg16_features():
    cfg = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")
    layers, c = [], 3
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(2, 2))
        else:
            layers += [nn.Conv2d(c, v, 3, padding=1), nn.ReLU(inplace=True)]
            c = v
    return nn.Sequential(*layers)


def main():
    tv = types.ModuleType("torchvision")
    models = types.ModuleType("torchvision.models")

    class _W:
        DEFAULT = None
    models.VGG16_Weights = _W
    models.vgg16 = lambda weights=None: types.SimpleNamespace(features=vgg16_features())
    tv.models = models
    sys.modules["torchvision"], sys.modules["torchvision.models"] = tv, models
    sys.path.insert(0, REF)
    import third_parties.lpips as ref_pkg
    from third_parties.lpips import LPIPS
    # the vendored lpips.py star-imports its package before the package has defined normalize_tensor (circular import):
    # hand it the function the package defines a few lines later (third_parties/lpips/__init__.py:13-15)
    import third_parties.lpips.lpips as ref_mod
    if not hasattr(ref_mod, "normalize_tensor"):
        ref_mod.normalize_tensor = ref_pkg.normalize_tensor
    m = LPIPS(net="vgg", pretrained=True, pnet_rand=True, verbose=False)
    with torch.no_grad():
        for salt, (name, p) in enumerate(sorted(m.net.named_parameters())):
            p.copy_(formula_weights(tuple(p.shape), salt))
    g = torch.Generator().manual_seed(20)
    x = torch.rand((4, 3, 32, 32), generator=g)
    y = (x + 0.15 * torch.randn((4, 3, 32, 32), generator=g)).clamp(0, 1)
    with torch.no_grad():
        val, per = m(x, y, retPerLayer=True, normalize=True)
    lin = {k: v.numpy() for k, v in m.state_dict().items() if k.startswith("lin")}
    np.savez_compressed(OUT, x=x.numpy(), y=y.numpy(), val=val.numpy(), per=np.stack([r.numpy() for r in per]),
                        param_order=np.array(sorted(n for n, _ in m.net.named_parameters())), **{k.replace(".", "__"): v for k, v in lin.items()})
    print("wrote", OUT, os.path.getsize(OUT), "bytes; val", val.reshape(-1).tolist())


if __name__ == "__main__":
    main()
