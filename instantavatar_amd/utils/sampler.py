"""Ray samplers on the device (drop-in names for instant_avatar/utils/sampler.py: `EdgeSampler`, `PatchSampler`;
`confs/sampler/{edge,patch}.yaml` re-point their `_target_` here).

    sampler.sample(mask, *args)  ->  [mask_s, *args_s]

as in the reference, but every tensor lives on the GPU and the pipeline np.where -> random choice -> gather runs as
HIP kernels without a host round trip (`ia_mask_edge`, `ia_nonzero_select`, gathers by index).  Random numbers are
uniform draws in [0,1) taken from torch's device generator, or injected (`draws=`) so that tests can feed the CPU
checker the same numbers.  The mapping from a uniform draw to an index is `floor(u * count)` -- the same
distribution as np.random.randint / np.random.choice, not the same stream.
"""
import torch

from .. import _lib


def _ws(nbytes, device):
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


def nonzero_select(mask2d, window, u, without_replacement=False):
    """(row, col) int32 [n] of `np.where(mask2d[y0:y1, x0:x1])` at the ranks derived from the uniform draws `u` [n]
    (coordinates relative to the window), plus the device-side count of nonzeros."""
    _lib.require_cuda(mask2d, u)
    H, W = mask2d.shape
    y0, y1, x0, x1 = window
    n = u.numel()
    L = _lib.lib()
    dev = mask2d.device
    row = torch.empty(n, dtype=torch.int32, device=dev)
    col = torch.empty(n, dtype=torch.int32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    ws = _ws(L.ia_nonzero_select_workspace_bytes(y1 - y0, n), dev)
    m = mask2d.float().contiguous()
    uu = u.float().contiguous()
    _lib.check(L.ia_nonzero_select(_lib.ptr(m), H, W, y0, y1, x0, x1, _lib.ptr(uu), n, int(without_replacement), _lib.ptr(row),
                                   _lib.ptr(col), _lib.ptr(count), _lib.ptr(ws), ws.numel(), _lib.stream()), "ia_nonzero_select")
    return row, col, count


class EdgeSampler:
    """instant_avatar/utils/sampler.py:5-46: num_mask pixels inside the mask, num_edge pixels of the band between the
    eroded and the dilated mask, the rest anywhere."""

    def __init__(self, num_sample, ratio_mask=0.6, ratio_edge=0.3, kernel_size=32):
        assert ratio_mask >= 0.0 and ratio_edge >= 0.0 and ratio_edge + ratio_mask <= 1.0
        self.kernel_size = int(kernel_size)
        self.num_mask = int(num_sample * ratio_mask)
        self.num_edge = int(num_sample * ratio_edge)
        self.num_rand = num_sample - self.num_mask - self.num_edge

    def edge_band(self, mask2d):
        """cv2.dilate(mask, ones(k,k)) - cv2.erode(mask, ones(k,k))  (:25-28) AS THE REFERENCE RUNS IT: `sample` flattens the
        mask first (:23), and OpenCV takes a 1-D array of length N as an N x 1 image -- the k x k box sees a single column,
        so the band is computed along the flattened (row-major) pixel index, window [-k//2, k-1-k//2], running across row
        ends.  The same kernel, called on an (H*W) x 1 image.  Returned in the mask's 2-D shape."""
        _lib.require_cuda(mask2d)
        H, W = mask2d.shape
        L = _lib.lib()
        m = mask2d.float().contiguous()
        edge = torch.empty_like(m)
        ws = _ws(L.ia_mask_edge_workspace_bytes(H * W, 1), m.device)
        _lib.check(L.ia_mask_edge(_lib.ptr(m), H * W, 1, self.kernel_size, _lib.ptr(edge), _lib.ptr(ws), ws.numel(), _lib.stream()),
                   "ia_mask_edge")
        return edge

    def sample_indices(self, mask2d, draws=None, generator=None):
        """flat pixel indices int32 [num_sample] in the reference's order: mask, edge, random (:33-41)."""
        H, W = mask2d.shape
        dev = mask2d.device
        n = self.num_mask + self.num_edge + self.num_rand
        if draws is None:
            draws = torch.rand(n, device=dev, generator=generator)
        u_m, u_e, u_r = draws[:self.num_mask], draws[self.num_mask:self.num_mask + self.num_edge], draws[self.num_mask + self.num_edge:n]
        r_m, c_m, _ = nonzero_select(mask2d, (0, H, 0, W), u_m)
        r_e, c_e, _ = nonzero_select(self.edge_band(mask2d), (0, H, 0, W), u_e)
        # an empty mask / an empty band (all-zero or all-one mask) selects nothing: ia_nonzero_select returns row = col = -1.
        # The reference raises there (np.random.randint(0, 0)); here those draws fall back to uniform pixels,
        # clamp(floor(u H W), max = H W - 1) -- like PatchSampler's empty-mask branch -- instead of handing negative pixel indices
        # to ia_sample_batch.  One launch (`ia_edge_indices`) for the index arithmetic, the fallbacks and the concatenation
        # (it was ~20 small torch launches of a 1.2 ms refine step).
        out = torch.empty(n, dtype=torch.int32, device=dev)
        d = draws[:n].float().contiguous()
        _lib.check(_lib.lib().ia_edge_indices(_lib.ptr(r_m), _lib.ptr(c_m), _lib.ptr(r_e), _lib.ptr(c_e), _lib.ptr(d), self.num_mask, self.num_edge,
                                              self.num_rand, H, W, _lib.ptr(out), _lib.stream()), "ia_edge_indices")
        return out

    def sample(self, mask, *args, draws=None, generator=None):
        mask2d = mask if mask.dim() == 2 else mask.reshape(mask.shape[0], -1)
        idx = self.sample_indices(mask2d, draws=draws, generator=generator).long()
        flat = mask2d.reshape(-1)
        out = [flat[idx]]
        for d in args:
            out.append(d.reshape(flat.numel(), -1)[idx])
        return out


class PatchSampler:
    """instant_avatar/utils/sampler.py:48-82: num_patch square patches; with probability ratio_mask their anchor pixels
    are drawn (without replacement) from the mask cropped by half a patch, else uniformly."""

    def __init__(self, num_patch=4, patch_size=20, ratio_mask=0.9, dilate=0):
        self.n = num_patch
        self.patch_size = patch_size
        self.p = ratio_mask
        self.dilate = dilate
        assert self.patch_size % 2 == 0, "patch size has to be even"

    def _candidates(self, mask2d):
        """the mask the anchors are drawn from: cv2.dilate(mask, ones(dilate, dilate)) > 0 when dilate > 0 (:62-65)"""
        if self.dilate <= 0:
            return mask2d
        _lib.require_cuda(mask2d)
        H, W = mask2d.shape
        L = _lib.lib()
        m = mask2d.float().contiguous()
        out = torch.empty_like(m)
        ws = _ws(L.ia_mask_edge_workspace_bytes(H, W), m.device)
        _lib.check(L.ia_mask_dilate(_lib.ptr(m), H, W, int(self.dilate), _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.stream()), "ia_mask_dilate")
        return (out > 0).float()

    def sample_corners(self, mask2d, draws=None, generator=None):
        """(row, col) int32 [num_patch] of the patches' top-left corners.  draws: [1 + 2 * num_patch] uniform numbers --
        the branch coin (:60), then the anchor draws (mask branch: the first num_patch; uniform branch: rows then columns)."""
        H, W = mask2d.shape
        dev = mask2d.device
        P = self.patch_size
        if draws is None:
            draws = torch.rand(1 + 2 * self.n, device=dev, generator=generator)
        draws = draws.float().contiguous()
        o = P // 2
        # both branches are evaluated on the device and blended by the coin: no host read of a random number.  The
        # uniform branch (np.random.randint(0, H - P) = floor(u * (H - P))) and the blend are one small kernel.
        r_m, c_m, count = nonzero_select(self._candidates(mask2d), (o, H - o, o, W - o), draws[1:1 + self.n], without_replacement=True)
        rows = torch.empty(self.n, dtype=torch.int32, device=dev)
        cols = torch.empty(self.n, dtype=torch.int32, device=dev)
        _lib.check(_lib.lib().ia_patch_corners(_lib.ptr(r_m), _lib.ptr(c_m), _lib.ptr(draws), self.n, H, W, P, float(self.p),
                                               _lib.ptr(rows), _lib.ptr(cols), _lib.stream()), "ia_patch_corners")
        return rows, cols

    def sample(self, mask, *args, draws=None, generator=None):
        mask2d = mask.reshape(mask.shape[0], mask.shape[1])
        rows, cols = self.sample_corners(mask2d, draws=draws, generator=generator)
        P = self.patch_size
        ar = torch.arange(P, device=mask.device)
        yy = (rows.long()[:, None, None] + ar[None, :, None]).expand(-1, P, P)
        xx = (cols.long()[:, None, None] + ar[None, None, :]).expand(-1, P, P)
        out = []
        for d in (mask, *args):
            p = d[yy, xx]
            if p.shape[-1] == 1 and p.dim() == 4:
                p = p.squeeze(-1)
            out.append(p)
        return out
