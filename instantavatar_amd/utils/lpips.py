"""LPIPS (Zhang et al. 2018, "The Unreasonable Effectiveness of Deep Features as a Perceptual Metric"), VGG-16 trunk,
version 0.1 -- the perceptual term of the reference's NGPLoss (instant_avatar/utils/loss.py:11,30-32, which builds
`third_parties/lpips.LPIPS(net="vgg", pretrained=True)`).

Not on the hot path: it is a convolutional network over the 4 rendered 32 x 32 patches of a training step and runs as
library convolutions (MIOpen through torch.nn.Conv2d).  What matters here is that a reference user can switch
over: parameter names and shapes equal the reference module's (`net.slice{1..5}.{i}.weight/bias`, `lin{0..4}.model.1.weight`,
buffers `scaling_layer.shift/scale`), so the reference's own weight files load unchanged:

  * the five 1x1 "lin" layers: `third_parties/lpips/weights/v0.1/vgg.pth` of a reference checkout (6.7 KB);
  * the VGG-16 feature trunk: torchvision's `vgg16(weights=DEFAULT).features.state_dict()` saved to a file (the
    reference lets torchvision download it; there is no network here, so the file has to be provided).

Without both files NGPLoss(w_lpips > 0) refuses to run rather than optimise against random features.

`LPIPS(net="alex")` is the metric of the reference's eval.py (eval.py:13-34: torchmetrics'
LearnedPerceptualImagePatchSimilarity(net_type="alex"), which wraps the same v0.1 network): AlexNet feature trunk
(torchvision `alexnet().features`, relu1..relu5 = 64 / 192 / 384 / 256 / 256 channels) + `weights/v0.1/alex.pth`.
"""
import torch
import torch.nn as nn

# torchvision's VGG-16 ("configuration D") feature stack up to relu5_3: output channels of the 3x3 convolutions, "M" =
# 2x2 max-pool.  Module indices follow torchvision's nn.Sequential (conv, relu alternate; a pool takes one index), which
# is what the slice boundaries (4, 9, 16, 23, 30) of the reference's trunk wrapper refer to.
_VGG16_D = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512)
_SLICE_ENDS = (4, 9, 16, 23, 30)
CHANNELS = (64, 128, 256, 512, 512)


def _vgg16_slices():
    layers, c_in = [], 3
    for v in _VGG16_D:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(c_in, v, kernel_size=3, padding=1), nn.ReLU(inplace=False)]
            c_in = v
    assert len(layers) == _SLICE_ENDS[-1]
    slices, start = [], 0
    for end in _SLICE_ENDS:
        seq = nn.Sequential()
        for i in range(start, end):
            seq.add_module(str(i), layers[i])       # keeps torchvision's indices as names: slice2.5.weight, ...
        slices.append(seq)
        start = end
    return slices


class VGG16Trunk(nn.Module):
    """relu1_2, relu2_2, relu3_3, relu4_3, relu5_3 of VGG-16."""

    def __init__(self):
        super().__init__()
        self.slice1, self.slice2, self.slice3, self.slice4, self.slice5 = _vgg16_slices()
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, x):
        outs = []
        for s in (self.slice1, self.slice2, self.slice3, self.slice4, self.slice5):
            x = s(x)
            outs.append(x)
        return outs

    def load_torchvision_features(self, state_dict):
        """`vgg16().features.state_dict()` (keys "0.weight", "2.bias", ...) -> the sliced layout."""
        own = {}
        for name, _ in self.named_parameters():
            own[name] = state_dict[name.split(".", 1)[1]]
        self.load_state_dict(own, strict=True)


# torchvision's AlexNet feature stack (indices = positions in its nn.Sequential) and the slice boundaries of the reference's
# trunk wrapper (third_parties/lpips/pretrained_networks.py:56-95): relu1 = [0,2), relu2 = [2,5), relu3 = [5,8), [8,10), [10,12)
_ALEX_SLICE_ENDS = (2, 5, 8, 10, 12)
ALEX_CHANNELS = (64, 192, 384, 256, 256)


def _alex_slices():
    layers = [nn.Conv2d(3, 64, kernel_size=11, stride=4, padding=2), nn.ReLU(inplace=False), nn.MaxPool2d(kernel_size=3, stride=2),
              nn.Conv2d(64, 192, kernel_size=5, padding=2), nn.ReLU(inplace=False), nn.MaxPool2d(kernel_size=3, stride=2),
              nn.Conv2d(192, 384, kernel_size=3, padding=1), nn.ReLU(inplace=False),
              nn.Conv2d(384, 256, kernel_size=3, padding=1), nn.ReLU(inplace=False),
              nn.Conv2d(256, 256, kernel_size=3, padding=1), nn.ReLU(inplace=False)]
    slices, start = [], 0
    for end in _ALEX_SLICE_ENDS:
        seq = nn.Sequential()
        for i in range(start, end):
            seq.add_module(str(i), layers[i])
        slices.append(seq)
        start = end
    return slices


class AlexTrunk(nn.Module):
    """relu1 .. relu5 of AlexNet."""

    def __init__(self):
        super().__init__()
        self.slice1, self.slice2, self.slice3, self.slice4, self.slice5 = _alex_slices()
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, x):
        outs = []
        for s in (self.slice1, self.slice2, self.slice3, self.slice4, self.slice5):
            x = s(x)
            outs.append(x)
        return outs

    load_torchvision_features = VGG16Trunk.load_torchvision_features


class _Lin(nn.Module):
    """1x1 convolution without bias; index 1 inside `model` (index 0 is the reference's dropout, inactive in eval mode)."""

    def __init__(self, c):
        super().__init__()
        self.model = nn.Sequential()
        self.model.add_module("1", nn.Conv2d(c, 1, 1, bias=False))

    def forward(self, x):
        return self.model(x)


class _Scaling(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("shift", torch.tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer("scale", torch.tensor([.458, .448, .450])[None, :, None, None])

    def forward(self, x):
        return (x - self.shift) / self.scale


class LPIPS(nn.Module):
    """d(x, y) = sum_l mean_hw( w_l . (f_l(x)/|f_l(x)| - f_l(y)/|f_l(y)|)^2 ), inputs NCHW in [0, 1] (normalize=True)."""

    def __init__(self, net="vgg"):
        super().__init__()
        if net not in ("vgg", "alex"):
            raise ValueError("LPIPS: net must be 'vgg' (the training loss) or 'alex' (eval.py's metric), got %r" % (net,))
        self.pnet_type = net
        self.scaling_layer = _Scaling()
        self.net = VGG16Trunk() if net == "vgg" else AlexTrunk()
        self.chns = CHANNELS if net == "vgg" else ALEX_CHANNELS
        for k, c in enumerate(self.chns):
            setattr(self, "lin%d" % k, _Lin(c))
        for p in self.parameters():
            p.requires_grad = False
        self.weights_loaded = {"trunk": False, "lin": False}
        self.eval()

    @property
    def lins(self):
        return [getattr(self, "lin%d" % k) for k in range(len(self.chns))]

    def load_lin_weights(self, path_or_state):
        sd = torch.load(path_or_state, map_location="cpu") if isinstance(path_or_state, (str, bytes)) or hasattr(path_or_state, "__fspath__") else path_or_state
        missing = [k for k in ("lin%d.model.1.weight" % i for i in range(len(CHANNELS))) if k not in sd]
        if missing:
            raise KeyError("LPIPS lin weights: missing %s (expected third_parties/lpips/weights/v0.1/%s.pth)" % (missing, self.pnet_type))
        self.load_state_dict({k: v for k, v in sd.items() if k.startswith("lin")}, strict=False)
        self.weights_loaded["lin"] = True
        return self

    def load_trunk_weights(self, path_or_state):
        sd = torch.load(path_or_state, map_location="cpu") if isinstance(path_or_state, (str, bytes)) or hasattr(path_or_state, "__fspath__") else path_or_state
        if any(k.startswith("features.") for k in sd):           # a whole torchvision vgg16 state dict
            sd = {k[len("features."):]: v for k, v in sd.items() if k.startswith("features.")}
        if any(k.startswith("net.") for k in sd):                # the sliced layout of a saved reference module
            self.load_state_dict({k: v for k, v in sd.items() if k.startswith("net.")}, strict=False)
        else:
            self.net.load_torchvision_features(sd)
        self.weights_loaded["trunk"] = True
        return self

    def forward(self, in0, in1, normalize=True, per_layer=False):
        if normalize:
            in0, in1 = 2 * in0 - 1, 2 * in1 - 1
        f0, f1 = self.net(self.scaling_layer(in0)), self.net(self.scaling_layer(in1))
        res = []
        for a, b, lin in zip(f0, f1, self.lins):
            a = a / (torch.sqrt(torch.sum(a ** 2, dim=1, keepdim=True)) + 1e-10)
            b = b / (torch.sqrt(torch.sum(b ** 2, dim=1, keepdim=True)) + 1e-10)
            res.append(lin((a - b) ** 2).mean([2, 3], keepdim=True))
        val = res[0]
        for r in res[1:]:
            val = val + r
        return (val, res) if per_layer else val
