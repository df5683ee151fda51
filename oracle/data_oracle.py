"""CPU checker (TEST INFRASTRUCTURE, numpy) of the data side of a training step -- a restatement
of instant_avatar/datasets/peoplesnapshot.py:12-25,99-151 and instant_avatar/utils/sampler.py:5-82 in which every
random number is an explicit input: `draws` are uniform numbers in [0,1) and an index is floor(u * count) computed in
float32, exactly the mapping the device samplers use (same distribution as np.random.randint / np.random.choice, not
numpy's stream).  cv2.erode / cv2.dilate with a k x k box are restated as window minima / maxima with the anchor at
(k//2, k//2) and out-of-image pixels ignored; tests/ cross-checks that against scipy.ndimage.
Only tests/ may import this module."""
import numpy as np


def make_rays(K, c2w, H, W):
    """peoplesnapshot.py:12-25"""
    x, y = np.meshgrid(np.arange(W), np.arange(H), indexing="xy")
    xy = np.stack([x, y, np.ones_like(x)], axis=-1).reshape(-1, 3).astype(np.float32)
    d_c = xy @ np.linalg.inv(K).T
    d_w = d_c @ c2w[:3, :3].T
    d_w = d_w / np.linalg.norm(d_w, axis=1, keepdims=True)
    o_w = np.tile(c2w[:3, 3], (len(d_w), 1))
    return o_w.reshape(H, W, 3).astype(np.float32), d_w.reshape(H, W, 3).astype(np.float32)


def _box_filter(a, k, fn, fill):
    """window extremum, window rows/cols [-k//2, k-1-k//2] around every pixel, pixels outside ignored (cv2 anchor + border)"""
    H, W = a.shape
    lo, hi = k // 2, k - 1 - k // 2
    pad = np.full((H + lo + hi, W + lo + hi), fill, a.dtype)
    pad[lo:lo + H, lo:lo + W] = a
    out = np.full((H, W), fill, a.dtype)
    for dy in range(k):
        for dx in range(k):
            out = fn(out, pad[dy:dy + H, dx:dx + W])
    return out


def erode(mask, k):
    return _box_filter(np.asarray(mask, np.float32), k, np.minimum, np.float32(np.inf))


def dilate(mask, k):
    return _box_filter(np.asarray(mask, np.float32), k, np.maximum, np.float32(-np.inf))


def _rank(u, count):
    r = np.floor(np.asarray(u, np.float32) * np.float32(count)).astype(np.int64)
    return np.minimum(r, count - 1)


def edge_sampler_indices(mask2d, draws, num_sample=4096, ratio_mask=0.6, ratio_edge=0.3, kernel_size=32):
    """EdgeSampler.sample's index computation (sampler.py:22-41) with explicit uniform draws [num_sample]."""
    num_mask, num_edge = int(num_sample * ratio_mask), int(num_sample * ratio_edge)
    mask = np.asarray(mask2d, np.float32).reshape(-1)
    # sampler.py:23-27: the mask is flattened BEFORE cv2.erode / cv2.dilate, and OpenCV takes a 1-D array of length N as
    # an N x 1 image: the k x k box only ever sees one column, i.e. the band is computed along the flattened (row-major)
    # index, window [-k//2, k-1-k//2], running across row ends.  Restated literally.
    col = mask.reshape(-1, 1)
    mask_e = (dilate(col, kernel_size) - erode(col, kernel_size)).reshape(-1)
    mask_loc, = np.where(mask)
    edge_loc, = np.where(mask_e)
    u = np.asarray(draws, np.float32)
    mask_idx = mask_loc[_rank(u[:num_mask], len(mask_loc))]
    edge_idx = edge_loc[_rank(u[num_mask:num_mask + num_edge], len(edge_loc))]
    rand_idx = _rank(u[num_mask + num_edge:num_sample], len(mask))
    return np.concatenate([mask_idx, edge_idx, rand_idx]).astype(np.int64)


def edge_sampler_sample(mask2d, args, draws, **kw):
    idx = edge_sampler_indices(mask2d, draws, **kw)
    mask = np.asarray(mask2d).reshape(-1)
    return [mask[idx]] + [np.asarray(d).reshape(len(mask), -1)[idx] for d in args]


def patch_sampler_corners(mask2d, draws, num_patch=4, patch_size=20, ratio_mask=0.9, dilate_k=0):
    """PatchSampler.sample's anchors (sampler.py:56-73).  draws [1 + 2*num_patch]: the coin of :60, then the anchor draws
    (mask branch: np.random.choice(replace=False) as sequential draws from the remaining candidates)."""
    u = np.asarray(draws, np.float32)
    H, W = mask2d.shape[:2]
    P = patch_size
    if u[0] < np.float32(ratio_mask):
        o = P // 2
        m = dilate(mask2d, dilate_k) > 0 if dilate_k > 0 else np.asarray(mask2d)   # sampler.py:62-65
        xs, ys = np.where(m[o:-o, o:-o] > 0)
        remaining = list(range(len(xs)))
        pick = []
        for i in range(num_patch):
            r = int(min(np.floor(u[1 + i] * np.float32(len(remaining))), len(remaining) - 1))
            pick.append(remaining.pop(r))
        pick = np.asarray(pick)
        return xs[pick], ys[pick]
    x = _rank(u[1:1 + num_patch], H - P)
    y = _rank(u[1 + num_patch:1 + 2 * num_patch], W - P)
    return x, y


def patch_sampler_sample(mask2d, args, draws, num_patch=4, patch_size=20, ratio_mask=0.9):
    x, y = patch_sampler_corners(mask2d, draws, num_patch, patch_size, ratio_mask)
    out = []
    for d in [mask2d, *args]:
        p = np.stack([np.asarray(d)[xi:xi + patch_size, yi:yi + patch_size] for xi, yi in zip(x, y)], axis=0)
        if p.shape[-1] == 1:
            p = p.squeeze(-1)
        out.append(p)
    return out


def getitem_train(img_u8, msk, rays_o, rays_d, smpl_params, idx, sample_fn, bg_full, near=None, far=None):
    """PeopleSnapshotDataset.__getitem__ for split == "train" (peoplesnapshot.py:99-151) without the file I/O / resize.
    bg_full [H,W,3]: the uniform background of :111; sample_fn(msk, img, rays_o, rays_d, bg) -> the sampler's output."""
    img = (np.asarray(img_u8)[..., :3] / 255).astype(np.float32)
    msk = np.asarray(msk).astype(np.float32)
    bg_color = np.asarray(bg_full, np.float32)
    img = img * msk[..., None] + (1 - msk[..., None]) * bg_color
    msk_s, img_s, ro, rd, bg_s = sample_fn(msk, img, rays_o, rays_d, bg_color)
    datum = {"rgb": img_s.astype(np.float32), "rays_o": ro, "rays_d": rd, "betas": smpl_params["betas"][0],
             "global_orient": smpl_params["global_orient"][idx], "body_pose": smpl_params["body_pose"][idx],
             "transl": smpl_params["transl"][idx], "alpha": msk_s, "bg_color": bg_s, "idx": idx}
    if near is not None and far is not None:
        datum["near"] = np.ones_like(rd[..., 0]) * near
        datum["far"] = np.ones_like(rd[..., 0]) * far
    else:
        dist = np.sqrt(np.square(smpl_params["transl"][idx]).sum(-1))
        datum["near"] = np.ones_like(rd[..., 0]) * (dist - 1)
        datum["far"] = np.ones_like(rd[..., 0]) * (dist + 1)
    return datum


def getitem_eval(img_u8, msk, rays_o, rays_d, smpl_params, idx, near=None, far=None):
    """PeopleSnapshotDataset.__getitem__ for split "val" / "test" (peoplesnapshot.py:112-125): the whole frame on a white
    background, everything flattened."""
    img = (np.asarray(img_u8)[..., :3] / 255).astype(np.float32)
    msk = np.asarray(msk).astype(np.float32)
    bg_color = np.ones_like(img).astype(np.float32)
    img = img * msk[..., None] + (1 - msk[..., None])
    rd = np.asarray(rays_d).reshape(-1, 3)
    datum = {"rgb": img.reshape(-1, 3).astype(np.float32), "rays_o": np.asarray(rays_o).reshape(-1, 3), "rays_d": rd,
             "betas": smpl_params["betas"][0], "global_orient": smpl_params["global_orient"][idx],
             "body_pose": smpl_params["body_pose"][idx], "transl": smpl_params["transl"][idx], "alpha": msk.reshape(-1),
             "bg_color": bg_color, "idx": idx}
    if near is not None and far is not None:
        datum["near"] = np.ones_like(rd[..., 0]) * near
        datum["far"] = np.ones_like(rd[..., 0]) * far
    else:
        dist = np.sqrt(np.square(smpl_params["transl"][idx]).sum(-1))
        datum["near"] = np.ones_like(rd[..., 0]) * (dist - 1)
        datum["far"] = np.ones_like(rd[..., 0]) * (dist + 1)
    return datum
