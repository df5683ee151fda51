"""f4 on the GPU (-m gpu): the device-side data path -- camera rays, Edge / Patch samplers, the per-step batch --
against oracle/data_oracle.py (numpy restatement of instant_avatar/datasets/peoplesnapshot.py:12-25,99-151 and
instant_avatar/utils/sampler.py:5-82) on the same uniform draws.  Integer results (pixel indices, patch anchors) are
bit-exact; rays within one float32 ulp (both sides evaluate in float64 and round once); composited colours exact."""
import numpy as np
import pytest
import torch

from instantavatar_amd.datasets.device_frames import DeviceFrames, make_rays
from instantavatar_amd.utils.sampler import EdgeSampler, PatchSampler, nonzero_select

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scene(H=135, W=120, N=3, seed=0):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    masks = np.stack([(((yy - H * 0.55) / (H * 0.33)) ** 2 + ((xx - W * (0.45 + 0.03 * i)) / (W * 0.18)) ** 2 < 1).astype(np.float32) for i in range(N)])
    masks[:, 10:14, 5:9] = 1.0                                    # a second blob
    imgs = rng.randint(0, 256, (N, H, W, 3)).astype(np.uint8)
    K = np.array([[2000.0 * H / 1080, 0, W / 2], [0, 2000.0 * H / 1080, H / 2], [0, 0, 1]])
    a = 0.3
    c2w = np.eye(4)
    c2w[:3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    c2w[:3, 3] = [0.1, -0.2, 0.3]
    smpl = dict(betas=rng.randn(1, 10).astype(np.float32), body_pose=rng.randn(N, 69).astype(np.float32) * 0.1,
                global_orient=rng.randn(N, 3).astype(np.float32) * 0.1, transl=(rng.randn(N, 3) * 0.1 + [0, 0.15, 5]).astype(np.float32))
    return imgs, masks, K, c2w, smpl


def test_make_rays_matches_reference_restatement():
    from oracle import data_oracle as do
    for (H, W) in ((135, 120), (512, 512)):
        _, _, K, c2w, _ = _scene(H, W, 1)
        o, d = make_rays(K, c2w, H, W, DEV)
        ro, rd = do.make_rays(K, c2w, H, W)
        assert np.array_equal(o.cpu().numpy(), ro)
        assert np.abs(d.cpu().numpy() - rd).max() <= 6e-8, np.abs(d.cpu().numpy() - rd).max()   # one float32 ulp below 1
        assert (d.cpu().numpy() != rd).mean() < 1e-3


def test_nonzero_select_is_np_where_at_rank():
    rng = np.random.RandomState(1)
    m = (rng.rand(97, 203) > 0.8).astype(np.float32)
    m[40] = 0                                                      # an empty row
    mt = torch.as_tensor(m, device=DEV)
    for window in ((0, 97, 0, 203), (5, 90, 7, 150)):
        y0, y1, x0, x1 = window
        rs, cs = np.where(m[y0:y1, x0:x1])
        u = rng.rand(500).astype(np.float32)
        u[:3] = [0.0, 0.99999994, 0.5]
        row, col, count = nonzero_select(mt, window, torch.as_tensor(u, device=DEV))
        assert int(count) == len(rs)
        rank = np.minimum(np.floor(u * np.float32(len(rs))).astype(np.int64), len(rs) - 1)
        assert np.array_equal(row.cpu().numpy(), rs[rank]) and np.array_equal(col.cpu().numpy(), cs[rank])
        # without replacement: sequential draws from the remaining candidates
        u2 = rng.rand(16).astype(np.float32)
        row, col, _ = nonzero_select(mt, window, torch.as_tensor(u2, device=DEV), without_replacement=True)
        remaining, pick = list(range(len(rs))), []
        for i in range(16):
            pick.append(remaining.pop(int(min(np.floor(u2[i] * np.float32(len(remaining))), len(remaining) - 1))))
        assert np.array_equal(row.cpu().numpy(), rs[pick]) and np.array_equal(col.cpu().numpy(), cs[pick])
        assert len(set(pick)) == 16
    row, col, count = nonzero_select(torch.zeros((8, 8), device=DEV), (0, 8, 0, 8), torch.rand(4, device=DEV))
    assert int(count) == 0 and (row == -1).all() and (col == -1).all()


def test_edge_sampler_matches_oracle():
    from oracle import data_oracle as do
    imgs, masks, K, c2w, smpl = _scene()
    rng = np.random.RandomState(2)
    s = EdgeSampler(num_sample=4096, ratio_mask=0.6, ratio_edge=0.3, kernel_size=16)      # confs/sampler/edge.yaml
    m = torch.as_tensor(masks[1], device=DEV)
    band = s.edge_band(m).cpu().numpy()
    col = masks[1].reshape(-1, 1)     # the reference's band is computed on the FLATTENED mask (sampler.py:23-27): an N x 1 image for OpenCV
    assert np.array_equal(band.reshape(-1), (do.dilate(col, 16) - do.erode(col, 16)).reshape(-1))
    draws = rng.rand(4096).astype(np.float32)
    idx = s.sample_indices(m, draws=torch.as_tensor(draws, device=DEV)).cpu().numpy()
    ref = do.edge_sampler_indices(masks[1], draws, 4096, 0.6, 0.3, 16)
    assert np.array_equal(idx, ref)
    img = rng.rand(135, 120, 3).astype(np.float32)
    out = s.sample(m, torch.as_tensor(img, device=DEV), draws=torch.as_tensor(draws, device=DEV))
    ref_out = do.edge_sampler_sample(masks[1], [img], draws, num_sample=4096, ratio_mask=0.6, ratio_edge=0.3, kernel_size=16)
    assert np.array_equal(out[0].cpu().numpy(), ref_out[0]) and np.array_equal(out[1].cpu().numpy(), ref_out[1])


def test_patch_sampler_matches_oracle_both_branches():
    from oracle import data_oracle as do
    imgs, masks, K, c2w, smpl = _scene()
    rng = np.random.RandomState(3)
    img = rng.rand(135, 120, 3).astype(np.float32)
    m = torch.as_tensor(masks[0], device=DEV)
    for coin, ratio in ((0.1, 1.0), (0.95, 0.9), (0.5, 0.9)):
        s = PatchSampler(num_patch=4, patch_size=32, ratio_mask=ratio)                      # confs/sampler/patch.yaml
        draws = np.r_[coin, rng.rand(8)].astype(np.float32)
        rows, cols = s.sample_corners(m, draws=torch.as_tensor(draws, device=DEV))
        x, y = do.patch_sampler_corners(masks[0], draws, 4, 32, ratio)
        assert np.array_equal(rows.cpu().numpy(), x) and np.array_equal(cols.cpu().numpy(), y), (coin, ratio)
        out = s.sample(m, torch.as_tensor(img, device=DEV), draws=torch.as_tensor(draws, device=DEV))
        ref = do.patch_sampler_sample(masks[0], [img], draws, 4, 32, ratio)
        assert out[0].shape == (4, 32, 32) and out[1].shape == (4, 32, 32, 3)
        assert np.array_equal(out[0].cpu().numpy(), ref[0]) and np.array_equal(out[1].cpu().numpy(), ref[1])
    # PatchSampler(dilate=8): anchors from the dilated mask (bash/run-neuman-demo.sh: sampler.dilate=8)
    s = PatchSampler(num_patch=4, patch_size=32, ratio_mask=1, dilate=8)
    draws = np.r_[0.0, rng.rand(8)].astype(np.float32)
    rows, cols = s.sample_corners(m, draws=torch.as_tensor(draws, device=DEV))
    x, y = do.patch_sampler_corners(masks[0], draws, 4, 32, 1, dilate_k=8)
    assert np.array_equal(rows.cpu().numpy(), x) and np.array_equal(cols.cpu().numpy(), y)
    assert np.array_equal(s._candidates(m).cpu().numpy() > 0, do.dilate(masks[0], 8) > 0)


@pytest.mark.parametrize("kind", ["edge", "patch"])
def test_device_batch_equals_reference_getitem(kind):
    """DeviceFrames.batch == PeopleSnapshotDataset.__getitem__ (train split) + the DataLoader's batch dimension."""
    from oracle import data_oracle as do
    imgs, masks, K, c2w, smpl = _scene()
    rng = np.random.RandomState(4)
    H, W = masks.shape[1:]
    if kind == "edge":
        sampler = EdgeSampler(num_sample=1024, ratio_mask=0.6, ratio_edge=0.3, kernel_size=16)
        draws = rng.rand(1024).astype(np.float32)
        fn = lambda msk, *args: do.edge_sampler_sample(msk, args, draws, num_sample=1024, ratio_mask=0.6, ratio_edge=0.3, kernel_size=16)
    else:
        sampler = PatchSampler(num_patch=4, patch_size=16, ratio_mask=1)
        draws = np.r_[0.3, rng.rand(8)].astype(np.float32)
        fn = lambda msk, *args: do.patch_sampler_sample(msk, args, draws, 4, 16, 1)
    frames = DeviceFrames.from_arrays(imgs, masks, K, c2w, smpl, sampler, DEV)
    ro, rd = do.make_rays(K, c2w, H, W)
    idx = 2
    bg_full = rng.rand(H, W, 3).astype(np.float32)
    ref = do.getitem_train(imgs[idx], masks[idx], ro, rd, smpl, idx, fn, bg_full)
    # the device draws the background for the sampled pixels only: hand it the reference's values at those pixels
    got = frames.batch(idx, draws=torch.as_tensor(draws, device=DEV),
                       bg_draws=torch.as_tensor(np.ascontiguousarray(ref["bg_color"]).reshape(-1, 3), device=DEV))
    for k in ("rgb", "alpha", "bg_color", "near", "far"):
        a, b = got[k][0].cpu().numpy(), np.asarray(ref[k], np.float32)
        assert a.shape == b.shape, (k, a.shape, b.shape)
        assert np.array_equal(a, b), (k, np.abs(a - b).max())
    for k in ("rays_o", "rays_d"):
        a, b = got[k][0].cpu().numpy(), ref[k]
        assert a.shape == b.shape and np.abs(a - b).max() <= 6e-8, k
    for k in ("betas", "global_orient", "body_pose", "transl"):
        assert np.array_equal(got[k][0].cpu().numpy(), ref[k]), k
    assert int(got["idx"][0]) == idx
    # out=: the next batch is written into the tensors of an earlier one (the static inputs of a captured training step)
    keep = {k: (v.data_ptr() if torch.is_tensor(v) and v.is_cuda else None) for k, v in got.items()}
    idx2 = 1
    ref2 = do.getitem_train(imgs[idx2], masks[idx2], ro, rd, smpl, idx2, fn, bg_full)
    again = frames.batch(idx2, draws=torch.as_tensor(draws, device=DEV),
                         bg_draws=torch.as_tensor(np.ascontiguousarray(ref2["bg_color"]).reshape(-1, 3), device=DEV), out=got)
    for k in ("rgb", "alpha", "bg_color", "near", "far", "rays_o", "rays_d", "betas", "global_orient", "body_pose", "transl"):
        assert again[k] is got[k] and again[k].data_ptr() == keep[k], k        # written in place, same tensor objects
    for k in ("rgb", "alpha", "bg_color", "near", "far", "betas", "global_orient", "body_pose", "transl"):
        assert np.array_equal(again[k][0].cpu().numpy(), np.asarray(ref2[k], np.float32)), k
    assert int(again["idx"][0]) == idx2


def test_device_eval_frame_equals_reference_getitem_val_split():
    """DeviceFrames.frame == PeopleSnapshotDataset.__getitem__ for the val / test split (whole frame, white background)."""
    from oracle import data_oracle as do
    imgs, masks, K, c2w, smpl = _scene()
    frames = DeviceFrames.from_arrays(imgs, masks, K, c2w, smpl, PatchSampler(num_patch=4, patch_size=16, ratio_mask=1), DEV)
    H, W = masks.shape[1:]
    ro, rd = do.make_rays(K, c2w, H, W)
    for idx in (0, 2):
        ref = do.getitem_eval(imgs[idx], masks[idx], ro, rd, smpl, idx)
        got = frames.frame(idx)
        for k in ("rgb", "alpha", "bg_color", "near", "far", "betas", "global_orient", "body_pose", "transl"):
            a, b = got[k][0].cpu().numpy(), np.asarray(ref[k], np.float32)
            assert a.shape == b.shape and np.array_equal(a, b), (k, a.shape, b.shape)
        for k in ("rays_o", "rays_d"):
            assert np.abs(got[k][0].cpu().numpy() - ref[k]).max() <= 6e-8, k
        assert int(got["idx"][0]) == idx


def test_device_batches_feed_a_training_step():
    """the device data path drives the real training step (plugins + kernels) end to end"""
    from instantavatar_amd import synthetic as syn
    from instantavatar_amd.pipeline import build_synthetic_model, make_batch
    from instantavatar_amd.training import NeRFLoss, configure_optimizer, training_step
    res = 128
    teacher, _, _ = build_synthetic_model(DEV, resolution=64)
    poses, tr = syn.procedural_pose_track(8)
    imgs, masks = [], []
    with torch.no_grad():
        for f in range(2):
            rgb, _, alpha, _ = teacher.render_image_fast(make_batch(DEV, res, poses[f], tr[f]), (res, res))
            imgs.append((rgb[0].clamp(0, 1) * 255).round().to(torch.uint8))
            masks.append((alpha[0] > 0.5).float())
    K = np.array([[2000.0 * res / 1080, 0, res / 2], [0, 2000.0 * res / 1080, res / 2], [0, 0, 1]])
    smpl = dict(betas=np.zeros((1, 10), np.float32), body_pose=poses[:2, 3:], global_orient=poses[:2, :3], transl=tr[:2])
    frames = DeviceFrames(torch.stack(imgs), torch.stack(masks), K, np.eye(4), smpl, EdgeSampler(num_sample=2048, kernel_size=8))
    model, _, _ = build_synthetic_model(DEV, resolution=64)
    model.net_coarse.reset_parameters()
    model.train()
    opt = configure_optimizer(model)
    loss_fn = NeRFLoss(dict(w_rgb=1.0, w_alpha=0.1, w_reg=0.1))
    first = last = None
    for it in range(30):
        losses = training_step(model, frames.batch(it % 2), opt, loss_fn)
        v = float(losses["mse_loss"].detach())
        first = v if first is None else first
        last = v
    assert np.isfinite(last) and last < first, (first, last)


def test_patch_sampler_empty_mask_falls_back_to_uniform_corners():
    """np.random.choice raises on an empty mask; the device sampler must not hand a negative pixel index to the gather:
    the corners then come from the uniform branch (always inside the image)."""
    H = W = 96
    m = torch.zeros((H, W), device=DEV)
    s = PatchSampler(num_patch=4, patch_size=32, ratio_mask=1)
    draws = torch.as_tensor(np.r_[0.0, np.linspace(0.05, 0.95, 8)].astype(np.float32), device=DEV)
    rows, cols = s.sample_corners(m, draws=draws)
    r, c = rows.cpu().numpy(), cols.cpu().numpy()
    assert (r >= 0).all() and (r < H - 32).all() and (c >= 0).all() and (c < W - 32).all()
    assert np.array_equal(r, np.floor(np.linspace(0.05, 0.95, 8)[:4].astype(np.float32) * np.float32(H - 32)).astype(np.int32))


def test_edge_sampler_on_empty_and_full_masks_stays_inside_the_image():
    """an all-zero (or all-one) mask has an empty mask set / an empty edge band; `ia_nonzero_select` then returns
    row = col = -1 and the flat index used to come out negative -- read by `ia_sample_batch` in front of the buffer.  Those
    draws now fall back to uniform pixels (the reference raises in np.random.randint(0, 0))."""
    from instantavatar_amd.utils.sampler import EdgeSampler
    H = W = 64
    s = EdgeSampler(num_sample=512, ratio_mask=0.6, ratio_edge=0.3, kernel_size=8)
    g = torch.Generator(device=DEV).manual_seed(0)
    for mask in (torch.zeros((H, W), device=DEV), torch.ones((H, W), device=DEV)):
        idx = s.sample_indices(mask, generator=g)
        assert idx.numel() == 512 and int(idx.min()) >= 0 and int(idx.max()) < H * W
    m = torch.zeros((H, W), device=DEV)
    m[20:40, 10:30] = 1
    idx = s.sample_indices(m, generator=g)
    assert int(idx.min()) >= 0 and int(idx.max()) < H * W
    assert bool((m.reshape(-1)[idx[:s.num_mask].long()] == 1).all())     # the mask share still comes from the mask
