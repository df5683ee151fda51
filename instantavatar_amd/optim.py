"""`FusedAdam`: the optimiser of DNeRFModel.configure_optimizers (DNeRF.py:32-59 -- `torch.optim.Adam`, three parameter
groups) whose step is ONE C-ABI call for all parameter tensors (`ia_adam_step`, csrc/ia_optim.hip): GradScaler's non-finite
check over every gradient (DNeRF.py:151-154: `self.scaler.step(optimizer)` skips the update), the Adam update, the refresh
of the fp16 copy of the field parameters that the kernels read, and (optionally) the gradient zero-fill for the next step.

State and state_dict have the layout of `torch.optim.Adam` -- per parameter `step` (float32 scalar tensor, kept on the
device like torch's capturable state), `exp_avg`, `exp_avg_sq`; per group lr / betas / eps / weight_decay / amsgrad /
maximize -- so an `optimizer_states` entry of a Lightning checkpoint written by the reference loads, and ours load there.
A learning-rate scheduler works as with torch's optimiser; `capturable_lr()` turns the group rates into device scalars so
that a step captured in a HIP graph sees later changes.  There is no CPU path: the step is the HIP kernel or an error.
"""
import ctypes as C

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    #: the step also zero-fills every gradient it consumed (saves the 52 MB fill of the next step).  Off by default: callers
    #: that look at `.grad` after a step (tests, gradient clipping experiments) see what the step consumed.
    fused_zero_grad = False

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, maximize=False):
        if weight_decay != 0 or amsgrad or maximize:
            raise NotImplementedError("FusedAdam: weight_decay / amsgrad / maximize are not used by the reference (DNeRF.py:46-50)")
        # (capturable / fused in the defaults: torch's load_state_dict then keeps `step` on the parameter's device)
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=True,
                        differentiable=False, fused=True)
        super().__init__(params, defaults)
        self._shadows = {}          # id(param) -> callable returning the fp16 copy (or None)
        self._ws = None
        self._found = None
        self.grads_zeroed = False   # every .grad buffer this optimiser owns is zero-filled (set by a fused_zero_grad step)

    def register_shadow(self, param, getter):
        """getter() -> fp16 tensor of param's shape that the step keeps equal to half(param), or None (no copy yet)"""
        self._shadows[id(param)] = getter

    def capturable_lr(self):
        """group learning rates as device scalars (read by the kernel at every launch / graph replay)"""
        for g in self.param_groups:
            if not torch.is_tensor(g["lr"]):
                dev = next((p.device for p in g["params"]), None) or next(p.device for gg in self.param_groups for p in gg["params"])
                g["lr"] = torch.tensor(float(g["lr"]), device=dev)
                if "initial_lr" in g and torch.is_tensor(g["initial_lr"]):
                    g["initial_lr"] = float(g["initial_lr"])

    def _state_of(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        if st["step"].device != p.device or st["step"].dtype != torch.float32:   # a state dict saved by a non-capturable Adam
            st["step"] = st["step"].to(device=p.device, dtype=torch.float32).reshape(())
        return st

    @torch.no_grad()
    def step(self, closure=None, skip_flag=None):
        """skip_flag: optional device scalar, non-zero = skip this update whatever the gradients hold.  Returns nothing
        (torch's contract); `found_inf` holds the device scalar written by the step (1.0 = skipped)."""
        if closure is not None:
            raise NotImplementedError("FusedAdam.step: no closure")
        items = []
        keep = []
        for g in self.param_groups:
            b1, b2 = g["betas"]
            for p in g["params"]:
                if p.grad is None:
                    continue
                _lib.require_cuda(p)
                if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous() or p.grad.dtype != torch.float32:
                    raise TypeError("FusedAdam: fp32 contiguous parameters and gradients only")
                st = self._state_of(p)
                sh = self._shadows.get(id(p))
                sh = sh() if sh is not None else None
                if sh is not None and (sh.dtype != torch.float16 or sh.numel() != p.numel() or not sh.is_contiguous()):
                    raise TypeError("FusedAdam: the fp16 copy must be a contiguous half tensor of the parameter's size")
                lr = g["lr"]
                t = _lib.AdamTensor(param=p.data_ptr(), grad=p.grad.data_ptr(), exp_avg=st["exp_avg"].data_ptr(), exp_avg_sq=st["exp_avg_sq"].data_ptr(),
                                    shadow=sh.data_ptr() if sh is not None else None, step=st["step"].data_ptr(),
                                    lr_dev=lr.data_ptr() if torch.is_tensor(lr) else None, lr=0.0 if torch.is_tensor(lr) else float(lr),
                                    beta1=float(b1), beta2=float(b2), eps=float(g["eps"]), numel=p.numel())
                items.append(t)
                keep.append((p.grad, sh, lr))
        if not items:
            return None
        dev = torch.device("cuda", torch.cuda.current_device())
        L = _lib.lib()
        if self._ws is None or self._ws.device != dev:
            self._ws = torch.zeros(int(L.ia_adam_workspace_bytes()), dtype=torch.uint8, device=dev)
            self._found = torch.zeros((), dtype=torch.float32, device=dev)
        if len(items) > _lib.IA_ADAM_MAX_TENSORS:
            raise NotImplementedError("FusedAdam: more than %d parameter tensors with gradients (the reference has at most 6)" % _lib.IA_ADAM_MAX_TENSORS)
        arr = (_lib.AdamTensor * len(items))(*items)
        skip = skip_flag.float().reshape(()).contiguous() if skip_flag is not None else None   # (named: alive until the launch is enqueued)
        _lib.check(L.ia_adam_step(arr, len(items), _lib.ptr(skip), _lib.ptr(self._found),
                                  1 if self.fused_zero_grad else 0, _lib.ptr(self._ws), self._ws.numel(), _lib.stream()), "ia_adam_step")
        self.grads_zeroed = bool(self.fused_zero_grad)
        return None

    def load_state_dict(self, state_dict):
        """also takes the state of a plain `torch.optim.Adam` (the reference's `optimizer_states`: `step` is a host tensor there and
        the groups say capturable / fused False): the groups keep this optimiser's flags, the counters move to the device"""
        super().load_state_dict(state_dict)
        for g in self.param_groups:
            g.update(capturable=True, fused=True, foreach=None, differentiable=False)
            for p in g["params"]:
                if p in self.state and "step" in self.state[p]:
                    self._state_of(p)

    @property
    def found_inf(self):
        return self._found

    def zero_grad(self, set_to_none=True):
        self.grads_zeroed = False
        return super().zero_grad(set_to_none=set_to_none)
