"""Image metrics of the reference's eval.py (eval.py:13-34, `Evaluator`): PSNR, SSIM and LPIPS(alex) between a rendered
frame and its ground truth -- the numbers the reference publishes for a trained avatar.

eval.py takes the three from torchmetrics (PeakSignalNoiseRatio(data_range=1), StructuralSimilarityIndexMeasure(data_range=1),
LearnedPerceptualImagePatchSimilarity(net_type="alex")).  torchmetrics is a third-party dependency that is neither vendored
under the reference nor installed here (it arrives unpinned with pytorch-lightning==1.5.7, install.sh:7), so:
  * PSNR and SSIM are restated from its published algorithm -- PARITY UNPINNED against torchmetrics itself; SSIM is
    cross-checked against an independent scipy implementation of Wang et al. 2004 with the same window
    (tests/test_cpu_oracle.py);
  * LPIPS is the v0.1 AlexNet network torchmetrics wraps, which the reference DOES vendor (third_parties/lpips): pinned to
    that module executing (tests/golden/lpips_alex_golden.npz).

Not on the hot path (library convolutions over a handful of frames at the end of a run); here so that a user of eval.py
finds its figures."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def psnr(pred, target, data_range=1.0):
    """10 log10(data_range^2 / MSE) over ALL elements of the call (torchmetrics' default: dim=None, base 10)."""
    mse = (pred.double() - target.double()).square().mean()
    return (10.0 / math.log(10.0)) * (2.0 * math.log(data_range) - torch.log(mse))


def _gaussian_window(kernel_size, sigma, channels, dtype, device):
    d = torch.arange((1 - kernel_size) / 2, (1 + kernel_size) / 2, 1, dtype=dtype, device=device)
    g = torch.exp(-(d / sigma) ** 2 / 2)
    g = (g / g.sum())[None]
    k2 = g.t() @ g                                            # [k, k]
    return k2.expand(channels, 1, kernel_size, kernel_size).contiguous()


def ssim(pred, target, data_range=1.0, kernel_size=11, sigma=1.5, k1=0.01, k2=0.03):
    """Structural similarity (Wang et al. 2004) as torchmetrics computes it: NCHW inputs, a normalised 11 x 11 Gaussian
    window (sigma 1.5) per channel, inputs reflect-padded by half a window, the statistics maps evaluated by a VALID
    convolution of the padded images and the outermost half window cropped off again, mean over channels and pixels of
    every image, then over the batch."""
    if pred.shape != target.shape or pred.dim() != 4:
        raise ValueError("ssim: NCHW tensors of equal shape expected, got %s and %s" % (tuple(pred.shape), tuple(target.shape)))
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    C = pred.shape[1]
    pad = (kernel_size - 1) // 2
    win = _gaussian_window(kernel_size, sigma, C, pred.dtype, pred.device)
    p = F.pad(pred, (pad, pad, pad, pad), mode="reflect")
    t = F.pad(target, (pad, pad, pad, pad), mode="reflect")
    stack = torch.cat([p, t, p * p, t * t, p * t])
    out = F.conv2d(stack, win, groups=C)
    mu_p, mu_t, pp, tt, pt = out.split(pred.shape[0])
    mu_pp, mu_tt, mu_pt = mu_p * mu_p, mu_t * mu_t, mu_p * mu_t
    s_pp, s_tt, s_pt = pp - mu_pp, tt - mu_tt, pt - mu_pt
    full = ((2 * mu_pt + c1) * (2 * s_pt + c2)) / ((mu_pp + mu_tt + c1) * (s_pp + s_tt + c2))
    full = full[..., pad:-pad, pad:-pad]
    return full.reshape(full.shape[0], -1).mean(-1).mean()


class Evaluator(nn.Module):
    """eval.py:13-34.  forward(rgb [N,H,W,3], rgb_gt [N,H,W,3]) -> {"psnr", "ssim", "lpips"}; the prediction is clamped to
    <= 1 (eval.py:26), inputs are cast to fp32 (`custom_fwd(cast_inputs=torch.float32)`).  LPIPS receives the [0, 1]
    images WITHOUT the [-1, 1] rescaling, as torchmetrics' default normalize=False does with what eval.py hands it.
    `lpips`: an utils.lpips.LPIPS(net="alex") with both weight files loaded; None -> the "lpips" entry is omitted (there is
    no silent stand-in for missing pretrained weights)."""

    def __init__(self, lpips=None):
        super().__init__()
        if lpips is not None and not all(lpips.weights_loaded.values()):
            raise ValueError("Evaluator: the LPIPS module has no pretrained weights loaded (%s)" % (lpips.weights_loaded,))
        self.lpips = lpips

    @torch.no_grad()
    def forward(self, rgb, rgb_gt):
        rgb = rgb.float().permute(0, 3, 1, 2).clamp(max=1.0)
        rgb_gt = rgb_gt.float().permute(0, 3, 1, 2)
        out = {"psnr": psnr(rgb, rgb_gt, 1.0).float(), "ssim": ssim(rgb, rgb_gt, 1.0)}
        if self.lpips is not None:
            out["lpips"] = self.lpips(rgb, rgb_gt, normalize=False).reshape(-1).mean()
        return out
