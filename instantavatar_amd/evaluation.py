"""The evaluation callers of the render path: DNeRFModel.validation_step / test_step (DNeRF.py:171-239) and what eval.py
does around them (eval.py:37-118: refine the SMPL parameters of the test frames with everything else frozen, render the
test frames, measure PSNR / SSIM / LPIPS on the written images).  Host logic only: every frame goes through
`AvatarModel.render_image_fast`, i.e. the HIP path.

Image files: the reference writes `cv2.imwrite(path, float_image * 255)` -- OpenCV converts a float image to 8 bits with
saturate_cast (round half to even, clamp to 0..255) and stores the array's channels as B, G, R; eval.py reads the files
back with cv2.imread + COLOR_BGR2RGB, i.e. it evaluates the channel-REVERSED 8-bit arrays (the reference's data set keeps
cv2.imread's channel order, peoplesnapshot.py:100-106, so that reversal is what hands true RGB to LPIPS).  `to_u8` /
`write_png_bgr` / `read_png_as_rgb` reproduce that round trip with PIL (cv2 is not installed here).
The error-map panel uses OpenCV's COLORMAP_JET; the table below is restated from its definition (clamp(1.5 - |4 v - k|) for k = 3, 2, 1
at v = i / 255) -- visualisation only, it never enters a metric, and it is UNPINNED (no cv2 to compare with)."""
import os

import numpy as np
import torch


def to_u8(x_times_255):
    """cv2.imwrite's float -> 8-bit conversion (saturate_cast<uchar>: round half to even, clamp)."""
    return torch.round(x_times_255).clamp_(0, 255).to(torch.uint8)


def _jet_lut_bgr():
    v = np.arange(256, dtype=np.float64) / 255.0
    ramp = lambda k: np.clip(1.5 - np.abs(4.0 * v - k), 0.0, 1.0)
    r, g, b = ramp(3.0), ramp(2.0), ramp(1.0)
    return np.rint(np.stack([b, g, r], -1) * 255.0).astype(np.uint8)


_JET = _jet_lut_bgr()


def jet_bgr(gray_u8):
    """cv2.applyColorMap(gray, cv2.COLORMAP_JET): uint8 [...] -> uint8 [..., 3] in B, G, R order."""
    lut = torch.as_tensor(_JET, device=gray_u8.device)
    return lut[gray_u8.long()]


@torch.no_grad()
def validation_step(model, batch, img_size):
    """DNeRF.py:171-188: render the whole frame, the three logged quantities."""
    rgb, depth, alpha, counter = model.render_image_fast(batch, img_size)
    rgb_gt = batch["rgb"].reshape(-1, *img_size, 3)
    return {"rgb_loss": (rgb - rgb_gt).square().mean(), "counter_avg": counter.mean(), "counter_max": counter.max(),
            "rgb": rgb, "alpha": alpha}


@torch.no_grad()
def test_step(model, batch, img_size):
    """DNeRF.py:226-239: [ground truth | rendering | error map] side by side, float in the model's channel order times
    1/255 -- what the reference hands to cv2.imwrite after multiplying by 255.  Returns the float panel [H, 3W, 3]."""
    rgb, *_ = model.render_image_fast(batch, img_size)
    rgb_gt = batch["rgb"].reshape(-1, *img_size, 3)
    err = (rgb - rgb_gt).square().sum(-1).sqrt()[0] / np.sqrt(3)
    errmap = jet_bgr((err * 255).clamp(0, 255).to(torch.uint8)).float()[None] / 255      # .astype(np.uint8): truncation
    return torch.cat([rgb_gt, rgb, errmap], dim=2)[0]


def write_png_bgr(path, panel_float):
    """cv2.imwrite(path, panel * 255) for a float [H, W, 3] panel whose channels are in the model's (B, G, R) order."""
    from PIL import Image
    u8 = to_u8(panel_float * 255).cpu().numpy()
    Image.fromarray(np.ascontiguousarray(u8[..., ::-1]), "RGB").save(path)


def read_png_as_rgb(path):
    """cv2.cvtColor(cv2.imread(path), cv2.COLOR_BGR2RGB) as a float tensor in [0, 1] (eval.py:98-100)."""
    from PIL import Image
    return torch.as_tensor(np.array(Image.open(path).convert("RGB"))).float() / 255.0


def freeze_all_but_smpl(model):
    """eval.py:70-73"""
    n = 0
    for k, p in model.named_parameters():
        if not k.startswith("SMPL_param"):
            p.requires_grad = False
        else:
            n += p.numel()
    return n


def evaluate_folder(folder, evaluator, device):
    """eval.py:97-118: every test/*.png is [gt | prediction | error map]; metrics of prediction against gt per image,
    averaged; writes nothing.  Returns ({"psnr", "ssim"[, "lpips"]}, n_images)."""
    names = sorted((f for f in os.listdir(folder) if f.endswith(".png")), key=lambda f: (len(f), f))
    if not names:
        raise FileNotFoundError("no test images in %s" % folder)
    rows = []
    for f in names:
        img = read_png_as_rgb(os.path.join(folder, f)).to(device)
        W = img.shape[1] // 3
        rows.append(evaluator(img[None, :, W:2 * W], img[None, :, :W]))
    return {k: float(torch.stack([r[k] for r in rows]).mean()) for k in rows[0]}, len(names)


def write_results(path, results):
    """results.txt of eval.py:107-118"""
    with open(path, "w") as f:
        f.write("PSNR: %.2f\n" % results["psnr"])
        f.write("SSIM: %.4f\n" % results["ssim"])
        if "lpips" in results:
            f.write("LPIPS: %.4f\n" % results["lpips"])
