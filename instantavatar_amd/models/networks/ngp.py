"""NeRFNGPNet plugin (drop-in for instant_avatar/models/networks/ngp.py:23).

The reference builds two tiny-cuda-nn modules (ngp.py:27-58); here the same
parameter tensors are plain torch Parameters with tcnn's names, shapes and
flat layouts, and the forward pass is the fused HIP kernel `ia_field_fwd`:

    encoder.params   fp32 [3072 + 2*n_entries]   = [W1 64x32 | W2 16x64 | hash table]
    color_net.params fp32 [6144]                 = [W1 64x16 | W2 64x64 | W3 16x64]
    center, scale    buffers (overwritten by initialize(bbox), ngp.py:64-71)

tcnn keeps fp32 master parameters and casts them to fp16 on every forward; the
fp16 shadow here is refreshed only when the master tensor changes
(`Tensor._version`).
"""
import ctypes as C

import numpy as np
import os

import torch
import torch.nn as nn

from ...deformers import _opt
from ... import _lib

EPS = 1e-3
N_LEVELS = 16
LOG2_HASHMAP = 19
BASE_RES = 16
PER_LEVEL_SCALE = 1.5
SIG_W1, SIG_W2 = 64 * 32, 16 * 64
COL_W1, COL_W2, COL_W3 = 64 * 16, 64 * 64, 16 * 64


class _TcnnParams(nn.Module):
    """Holder with a single flat fp32 `params` tensor, as tcnn's torch Module."""

    def __init__(self, n):
        super().__init__()
        self.params = nn.Parameter(torch.zeros(n, dtype=torch.float32))


class NeRFNGPNet(nn.Module):
    def __init__(self, opt, n_levels=N_LEVELS, log2_hashmap_size=LOG2_HASHMAP, level3_res=None):
        super().__init__()
        self.n_levels = n_levels
        self.log2_T = log2_hashmap_size
        # level3_res (54 | 55 | None = host libm / IA_TCNN_LEVEL3_RES): see _lib.apply_level3_override
        self.hash_desc = _lib.make_hash_desc(n_levels, log2_hashmap_size, BASE_RES, PER_LEVEL_SCALE, level3_res=level3_res)
        self.n_entries = int(self.hash_desc.offset[n_levels])
        self.sig_w1_size = 64 * 2 * n_levels
        self.encoder = _TcnnParams(self.sig_w1_size + SIG_W2 + 2 * self.n_entries)
        self.color_net = _TcnnParams(COL_W1 + COL_W2 + COL_W3)
        self.register_buffer("center", torch.tensor(list(_opt.get(opt, "center", [0, -0.3, 0])), dtype=torch.float32))
        self.register_buffer("scale", torch.tensor(list(_opt.get(opt, "scale", [2.5, 2.5, 2.5])), dtype=torch.float32))
        self.opt = opt
        self._half = None
        self._half_key = None
        self._desc = None
        self.reset_parameters()

    def clone_shared(self):
        """A second handle on the SAME weights (the two `_TcnnParams` modules and the centre / scale buffers are shared by
        reference) with its own kernel-side scratch: fp16 shadow cache, MFMA fragment image, sharded-encoding planes and
        C descriptor.  Lets two frames be in flight on two streams (pipeline.PipelinedRenderer) without sharing scratch."""
        other = NeRFNGPNet.__new__(NeRFNGPNet)
        nn.Module.__init__(other)
        other.n_levels, other.log2_T, other.hash_desc = self.n_levels, self.log2_T, self.hash_desc
        other.n_entries, other.sig_w1_size = self.n_entries, self.sig_w1_size
        other.encoder, other.color_net = self.encoder, self.color_net
        other.register_buffer("center", self.center)
        other.register_buffer("scale", self.scale)
        other.opt = self.opt
        other._half = other._half_key = other._desc = None
        if hasattr(self, "bbox"):
            other.bbox = self.bbox
        return other

    def reset_parameters(self, seed=1337):
        """tcnn defaults: hash table U(-1e-4, 1e-4), MLP weights Xavier-uniform."""
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            p = self.encoder.params
            o = 0
            for out_f, in_f in ((64, 2 * self.n_levels), (16, 64)):
                a = (6.0 / (out_f + in_f)) ** 0.5
                p[o:o + out_f * in_f] = (torch.rand(out_f * in_f, generator=g) * 2 - 1) * a
                o += out_f * in_f
            p[o:] = (torch.rand(p.numel() - o, generator=g) * 2 - 1) * 1e-4
            q = self.color_net.params
            o = 0
            for out_f, in_f in ((64, 16), (64, 64), (16, 64)):
                a = (6.0 / (out_f + in_f)) ** 0.5
                q[o:o + out_f * in_f] = (torch.rand(out_f * in_f, generator=g) * 2 - 1) * a
                o += out_f * in_f

    def load_field_dict(self, fp):
        """Load weights produced by instantavatar_amd.synthetic.make_field (fp16 numpy)."""
        with torch.no_grad():
            dev = self.encoder.params.device
            t = lambda k: torch.from_numpy(np.asarray(fp[k], dtype=np.float32).reshape(-1)).to(dev)
            self.encoder.params.copy_(torch.cat([t("sig_w1"), t("sig_w2"), t("table")]))
            self.color_net.params.copy_(torch.cat([t("col_w1"), t("col_w2"), t("col_w3")]))
            self.center = torch.as_tensor(fp["center"], dtype=torch.float32, device=dev)
            self.scale = torch.as_tensor(fp["scale"], dtype=torch.float32, device=dev)
        self._desc = None

    @staticmethod
    def tcnn_encoder_sizes(n_levels=N_LEVELS, log2_hashmap_size=LOG2_HASHMAP):
        """{level-3 resolution: numel of `encoder.params`} for the two layouts tcnn v1.6 can produce (see
        _lib.apply_level3_override): [W1 64x2L | W2 16x64 | grid]."""
        out = {}
        for r3 in (54, 55):
            hd = _lib.make_hash_desc(n_levels, log2_hashmap_size, BASE_RES, PER_LEVEL_SCALE, level3_res=r3)
            out[r3] = 64 * 2 * n_levels + SIG_W2 + 2 * int(hd.offset[n_levels])
        return out

    def adopt_tcnn_layout(self, encoder_numel):
        """Decide the one open layout question of tcnn v1.6 -- level-3 resolution 54 or 55 (it depends on the last bit of the
        build's exp2f, _lib.apply_level3_override) -- from the SIZE of a real `encoder.params` vector, and switch this module to
        it (level table, parameter vector, kernel descriptor).  Returns the resolution adopted; raises when the size fits
        neither layout.  With the first real checkpoint this pins SURVEY row a10 in one run (see `self_check`)."""
        sizes = self.tcnn_encoder_sizes(self.n_levels, self.log2_T)
        fit = [r for r, n in sizes.items() if n == int(encoder_numel)]
        if not fit:
            raise ValueError("encoder.params with %d elements fits neither tcnn layout (level-3 resolution 54: %d, 55: %d elements)"
                             % (encoder_numel, sizes[54], sizes[55]))
        r3 = fit[0] if self.n_levels > 3 else int(self.hash_desc.res[min(3, self.n_levels - 1)])
        if self.n_levels > 3 and int(self.hash_desc.res[3]) != r3:
            dev = self.encoder.params.device
            self.hash_desc = _lib.make_hash_desc(self.n_levels, self.log2_T, BASE_RES, PER_LEVEL_SCALE, level3_res=r3)
            self.n_entries = int(self.hash_desc.offset[self.n_levels])
            # resized IN PLACE: the Parameter object stays the one an optimiser (built before the load: drivers/train.py
            # --resume), a DDP-style reducer or a captured graph's owner already holds -- a new module here would leave them
            # with an orphan and the live table would never be updated.  Stale per-parameter optimiser state of
            # the old shape is the caller's to drop (load_checkpoint does, before it restores the checkpoint's own).
            p = self.encoder.params
            p.data = torch.zeros(self.sig_w1_size + SIG_W2 + 2 * self.n_entries, dtype=torch.float32, device=dev)
            p.grad = None
            self._half = self._half_key = self._desc = None
            self.reset_parameters()
        self.tcnn_level3_res = r3
        return r3

    @torch.no_grad()
    def self_check(self, n=8192, seed=0):
        """Sanity of a freshly loaded parameter set on the device: the hash-grid features of `n` random points of the unit
        cube must be finite and no level may be constant (a layout that reads a checkpoint shifted by even one level offset
        shows up as levels of pure initialisation noise or zeros), sigma / rgb must be finite.  Returns a report dict;
        raises ValueError on a failed check."""
        dev = self.encoder.params.device
        g = torch.Generator(device=dev).manual_seed(seed)
        x = (torch.rand((n, 3), device=dev, generator=g) - 0.5) * self.scale.to(dev) + self.center.to(dev)
        feat = self.encode(x).float().reshape(n, self.n_levels, 2)
        rgb, sigma = self.forward(x)
        std = feat.std(dim=0).amax(dim=1)
        rep = {"level3_res": int(self.hash_desc.res[min(3, self.n_levels - 1)]), "finite": bool(torch.isfinite(feat).all() and torch.isfinite(sigma).all()
                                                                                               and torch.isfinite(rgb).all()),
               "feature_std_per_level": [float(v) for v in std], "constant_levels": [int(i) for i in (std == 0).nonzero().reshape(-1)],
               "encoder_numel": int(self.encoder.params.numel())}
        if not rep["finite"] or rep["constant_levels"]:
            raise ValueError("NeRFNGPNet.self_check failed: %s" % rep)
        return rep

    def load_tcnn_params(self, encoder_params, color_params):
        """Load the reference's two flat tcnn parameter vectors (state-dict entries `net_coarse.encoder.params`,
        `net_coarse.color_net.params`, ngp.py:27-58).  Sizes are checked against this module's level table; a
        vector with the OTHER level-3 layout is rejected with the size it has, the size expected and the switch to
        flip -- never loaded shifted."""
        enc, col = torch.as_tensor(encoder_params).reshape(-1), torch.as_tensor(color_params).reshape(-1)
        sizes = self.tcnn_encoder_sizes(self.n_levels, self.log2_T)
        own = self.encoder.params.numel()
        if enc.numel() != own:
            other = [r for r, n in sizes.items() if n == enc.numel()]
            hint = (" -- that is the layout with a level-3 resolution of %d (this module was built for %d): construct "
                    "NeRFNGPNet(..., level3_res=%d) or set IA_TCNN_LEVEL3_RES=%d" % (other[0], int(self.hash_desc.res[3]), other[0], other[0])
                    ) if other and self.n_levels > 3 else ""
            raise ValueError("encoder.params has %d elements, expected %d (= 64x%d + 16x64 + 2 x %d grid entries)%s"
                             % (enc.numel(), own, 2 * self.n_levels, self.n_entries, hint))
        if col.numel() != self.color_net.params.numel():
            raise ValueError("color_net.params has %d elements, expected %d (64x16 + 64x64 + 16x64)" % (col.numel(), self.color_net.params.numel()))
        with torch.no_grad():
            self.encoder.params.copy_(enc.to(self.encoder.params))
            self.color_net.params.copy_(col.to(self.color_net.params))
        self.mark_updated()

    def initialize(self, bbox):
        """ngp.py:64-71"""
        if hasattr(self, "bbox"):
            return
        self.center = (bbox[0] + bbox[1]) / 2
        self.scale = (bbox[1] - bbox[0])
        self.bbox = bbox
        self._desc = None

    # -- fp16 shadow + C descriptor ------------------------------------------
    def mark_updated(self, shadow_fresh=False):
        """Mark the fp16 shadow stale.  Call after an optimizer step: fused / foreach
        optimizers update parameters without bumping Tensor._version.
        shadow_fresh: the step itself wrote the fp16 copy (optim.FusedAdam -> `half_shadow`): only the MFMA fragment image
        (a re-arrangement of the fp16 MLP weights) is rebuilt on next use, the two fp32 -> fp16 casts are skipped."""
        if shadow_fresh and self._half is not None and getattr(self, "_dirty", False) is not True:
            self._dirty = "frags"
        else:
            self._dirty = True

    def half_shadow(self, which):
        """the fp16 copy of encoder.params (0) / color_net.params (1) that the kernels read, or None before its first use"""
        h = self._half
        if h is None or h[0].device != self.encoder.params.device or h[which].numel() != (self.encoder, self.color_net)[which].params.numel():
            return None
        return h[which]

    def _half_params(self):
        """fp16 shadow of the two parameter vectors.  It is refreshed IN PLACE (same device pointers) whenever the master
        weights changed, and so is the MFMA fragment image: a captured HIP graph (pipeline.GraphedRenderer) has these
        pointers baked in and sees new weights after the next eager `field_desc()` / `refresh()` call."""
        key = (self.encoder.params._version, self.color_net.params._version, self.encoder.params.data_ptr())
        dirty = getattr(self, "_dirty", False)
        stale = bool(dirty) or self._half_key != key
        if self._half is None or self._half[0].device != self.encoder.params.device:
            self._half = (self.encoder.params.detach().to(torch.float16).contiguous(),
                          self.color_net.params.detach().to(torch.float16).contiguous())
            self._desc = None
        elif stale:
            if dirty != "frags" or self._half_key != key:   # ("frags": the optimiser step wrote the fp16 copy itself)
                with torch.no_grad():
                    self._half[0].copy_(self.encoder.params.detach())
                    self._half[1].copy_(self.color_net.params.detach())
            if self._desc is not None and getattr(self, "_frags", None) is not None:
                _lib.check(_lib.lib().ia_field_prepare(C.byref(self._desc), _lib.ptr(self._frags), _lib.stream()), "ia_field_prepare")
        self._half_key = key
        self._dirty = False
        return self._half

    def refresh(self):
        """bring the kernel-side copies (fp16 shadow, MFMA fragments) up to date now, on the current stream"""
        self.mark_updated()
        return self.field_desc()

    #: upper bound of the XCD-sharded encoding scratch (64 B per sample); larger calls use the
    #: single fused kernel.  0 disables the sharded path.
    max_encode_workspace_bytes = 2 << 30

    #: `ia_field.enc_split`: share (tiles of every four) of the sharded encoder's second level group that XCDs 0-3 take.  The samples
    #: this network is queried with on the path -- a frame's candidates, the occupancy probes, a training batch -- are spatially
    #: coherent: 3 (605 -> 612 frames/s); `sample_coherence(False)` for arbitrary points (2: 17 % faster on uniformly random ones)
    enc_split = int(os.environ.get("IA_ENC_SPLIT", "3"))    # (the environment variable: A/B runs)
    #: the same hint for the training forward (`training.field_autograd`): the candidates of a patch / uniform-ray batch are fewer and
    #: less regular than a frame's -- 2 measured 0.5 % ahead of 3 on the patch workload (573 vs 570 it/s), equal on the others
    enc_split_train = int(os.environ.get("IA_ENC_SPLIT_TRAIN", "2"))

    def sample_coherence(self, coherent=True):
        """Tell the sharded encoder whether the samples of the following queries are spatially coherent (see `enc_split`);
        results do not depend on it."""
        self.enc_split = 3 if coherent else 2
        if getattr(self, "_desc", None) is not None:
            self._desc.enc_split = int(self.enc_split)

    def _reserve_encode_workspace(self, n_samples, device):
        """Level-plane scratch of the XCD-sharded encoding (`ia_field.enc_ws`): grown on demand,
        never shrunk (so a captured HIP graph keeps valid pointers once sizes have settled)."""
        have = getattr(self, "_enc_ws_samples", 0)
        if n_samples <= have or n_samples * 4 * self.n_levels > self.max_encode_workspace_bytes:
            return
        n = (int(n_samples) + 1023) // 1024 * 1024
        self._enc_ws = torch.empty((self.n_levels, n), dtype=torch.int32, device=device)
        self._enc_ws_samples = n
        if self._desc is not None:
            self._desc.enc_ws, self._desc.enc_ws_samples = self._enc_ws.data_ptr(), n

    def _center_scale_host(self):
        """Host copies of center / scale for the C descriptor.  Read back only when the tensors changed
        (initialize / checkpoint load): the descriptor is rebuilt after every optimizer step and a
        device read there would synchronise the training loop once per step."""
        key = (self.center.data_ptr(), self.center._version, self.scale.data_ptr(), self.scale._version)
        if getattr(self, "_cs_key", None) != key:
            self._cs_host = (self.center.detach().float().cpu().tolist(), self.scale.detach().float().cpu().tolist())
            self._cs_key = key
        return self._cs_host

    def field_desc(self, max_samples=0):
        """C descriptor of the field.  `max_samples`: largest sample count (capacity) of the
        call it is built for; reserves the sharded-encoding scratch for it."""
        enc, col = self._half_params()
        if max_samples:
            self._reserve_encode_workspace(max_samples, enc.device)
        if self._desc is not None:
            # centre / scale are host copies inside the descriptor: a checkpoint load or a broadcast that overwrote the
            # two buffers in place (only `mark_updated()` follows those) must reach it too -- keyed on the tensors'
            # versions, so the device is read back only when they really changed.  (Kernel arguments of an already
            # captured HIP graph keep the old values: re-capture after loading a checkpoint with another bbox.)
            cs = self._center_scale_host()
            if self._cs_key != getattr(self, "_desc_cs_key", None):
                self._desc.center[:], self._desc.scale[:] = cs
                self._desc_cs_key = self._cs_key
        if self._desc is None:
            f = _lib.Field()
            f.center[:], f.scale[:] = self._center_scale_host()
            self._desc_cs_key = self._cs_key
            f.hash = self.hash_desc
            e, c = enc.data_ptr(), col.data_ptr()
            f.sig_w1 = e
            f.sig_w2 = e + 2 * self.sig_w1_size
            f.table = e + 2 * (self.sig_w1_size + SIG_W2)
            f.col_w1 = c
            f.col_w2 = c + 2 * COL_W1
            f.col_w3 = c + 2 * (COL_W1 + COL_W2)
            f.mlp_frags = None
            # MFMA weight fragments: rebuilt only when the fp16 shadow changed
            L = _lib.lib()
            self._frags = torch.empty(L.ia_field_frags_bytes() // 2, dtype=torch.float16, device=enc.device)
            _lib.check(L.ia_field_prepare(C.byref(f), _lib.ptr(self._frags), _lib.stream()), "ia_field_prepare")
            f.mlp_frags = self._frags.data_ptr()
            if getattr(self, "_enc_ws_samples", 0):
                f.enc_ws, f.enc_ws_samples = self._enc_ws.data_ptr(), self._enc_ws_samples
            f.enc_split = int(self.enc_split)
            self._desc = f
        return self._desc

    def forward(self, x, d=None, cond=None):
        """ngp.py:73-83 -> (color [V,3] fp32, sigma [V] fp32).
        The reference asserts the input range with a host sync (ngp.py:76); the
        kernel clamps like ngp.py:77 and never synchronises."""
        _lib.require_cuda(x)
        if torch.is_grad_enabled() and (self.encoder.params.requires_grad or x.requires_grad):
            from ...training import field_autograd
            return field_autograd(self, x)
        xc = x.detach().reshape(-1, 3).float().contiguous()
        V = xc.shape[0]
        rgb = torch.empty((V, 3), device=x.device)
        sigma = torch.empty(V, device=x.device)
        _lib.check(_lib.lib().ia_field_fwd(_lib.ptr(xc), V, None, C.byref(self.field_desc(V)), _lib.ptr(rgb),
                                           _lib.ptr(sigma), _lib.stream()), "ia_field_fwd")
        return rgb, sigma

    def encode(self, x):
        """hash-grid features only (fp16 [V, 2*n_levels]); the roofline kernel in isolation."""
        _lib.require_cuda(x)
        xc = x.detach().reshape(-1, 3).float().contiguous()
        feat = torch.empty((xc.shape[0], 2 * self.n_levels), device=x.device, dtype=torch.float16)
        _lib.check(_lib.lib().ia_hashgrid_fwd(_lib.ptr(xc), xc.shape[0], C.byref(self.field_desc()), _lib.ptr(feat),
                                              _lib.stream()), "ia_hashgrid_fwd")
        return feat

    def encode_planes(self, x):
        """hash-grid features from the XCD-sharded kernel: int32 [n_levels, V] of packed half2
        (level-major planes, what the fused field kernel consumes)."""
        _lib.require_cuda(x)
        xc = x.detach().reshape(-1, 3).float().contiguous()
        V = xc.shape[0]
        planes = torch.empty((self.n_levels, V), device=x.device, dtype=torch.int32)
        _lib.check(_lib.lib().ia_hashgrid_fwd_planes(_lib.ptr(xc), V, C.byref(self.field_desc()), _lib.ptr(planes), V,
                                                     _lib.stream()), "ia_hashgrid_fwd_planes")
        return planes
