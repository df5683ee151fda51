"""GPU tests (-m gpu) of the fused optimiser step `ia_adam_step` (csrc/ia_optim.hip, optim.FusedAdam): GradScaler's non-finite
check + torch.optim.Adam + fp16 copy + gradient zero-fill of DNeRFModel.training_step (DNeRF.py:46-50, :151-159) in one call,
against oracle.adam_step (pinned to torch.optim.Adam on the CPU: test_adam_oracle_matches_torch_adam_...)."""
import numpy as np
import pytest
import torch

from instantavatar_amd.optim import FusedAdam

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(oracle, zero_grad, lr_tensor, n_steps=20, bad_steps=(7,), extra_skip_steps=(11,)):
    rs = np.random.RandomState(1)
    shapes, lrs = [(1 << 20,), (40003,), (64, 16), (7, 72), (5,)], [1e-2, 1e-2, 1e-2, 1e-5, 1e-5]
    init = [rs.randn(*s).astype(np.float32) * 0.1 for s in shapes]
    P = [torch.nn.Parameter(torch.as_tensor(a.copy(), device=DEV)) for a in init]
    opt = FusedAdam([{"params": P[:2]}, {"params": [P[2]]}, {"params": P[3:], "lr": 1e-5}], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    opt.fused_zero_grad = zero_grad
    shadow = torch.zeros(shapes[0], dtype=torch.float16, device=DEV)
    opt.register_shadow(P[0], lambda: shadow)
    if lr_tensor:
        opt.capturable_lr()       # float32 device scalars: the rate the kernel reads is float32(lr)
        lrs = [float(np.float32(v)) for v in lrs]
    p = [a.copy() for a in init]
    st = [dict(step=0.0, exp_avg=np.zeros(s, np.float32), exp_avg_sq=np.zeros(s, np.float32)) for s in shapes]
    for k in range(n_steps):
        g = [(rs.randn(*s) * 10.0 ** rs.uniform(-7, 0)).astype(np.float32) for s in shapes]
        for a in g:
            a.reshape(-1)[::5] = 0
        if k in bad_steps:
            g[2].reshape(-1)[3] = np.inf if k % 2 else np.nan
        extra = k in extra_skip_steps
        if k == 15:                       # an lr scheduler between steps (LambdaLR assigns floats or fills the tensor)
            for grp in opt.param_groups:
                if torch.is_tensor(grp["lr"]):
                    grp["lr"].fill_(float(grp["lr"]) * 0.5)
                else:
                    grp["lr"] *= 0.5
            lrs = [v * 0.5 for v in lrs]
        for q, a in zip(P, g):
            if q.grad is None or not zero_grad:
                q.grad = torch.as_tensor(a.copy(), device=DEV)
            else:
                assert float(q.grad.abs().max()) == 0.0       # left zero-filled by the previous step
                q.grad.copy_(torch.as_tensor(a, device=DEV))
        opt.step(skip_flag=torch.tensor(1.0 if extra else 0.0, device=DEV))
        found = oracle.adam_step(p, [a.copy() for a in g], st, lrs, skip=extra)
        assert found == (k in bad_steps or extra) and float(opt.found_inf) == float(found), (k, found, float(opt.found_inf))
        if not zero_grad:
            assert all(torch.equal(q.grad.cpu(), torch.as_tensor(a)) or not np.isfinite(a).all() for q, a in zip(P, g))
    return P, opt, shadow, p, st, lrs


@pytest.mark.parametrize("zero_grad,lr_tensor", [(False, False), (True, True)])
def test_fused_adam_step_matches_oracle(oracle, zero_grad, lr_tensor):
    P, opt, shadow, p, st, lrs = _run(oracle, zero_grad, lr_tensor)
    n_skipped = 2
    for q, a, s in zip(P, p, st):
        os_ = opt.state[q]
        assert float(os_["step"]) == s["step"] == 20 - n_skipped
        m, v, pp = os_["exp_avg"].cpu().numpy(), os_["exp_avg_sq"].cpu().numpy(), q.detach().cpu().numpy()
        assert np.array_equal(m, s["exp_avg"]) and np.array_equal(v, s["exp_avg_sq"])     # no transcendental involved: bit-equal
        d = np.abs(pp - a)
        print("tensor %s: parameters bit-equal on %.6f of the elements, max |diff| %.3e" % (tuple(a.shape), float((d == 0).mean()), float(d.max())))
        assert np.array_equal(pp, a)
    # the fp16 copy written in the same pass equals a cast of the updated master copy
    assert torch.equal(shadow, P[0].detach().to(torch.float16))
    if zero_grad:
        assert all(float(q.grad.abs().max()) == 0.0 for q in P)


def test_fused_adam_state_dict_round_trips_with_torch_adam():
    """`optimizer_states` of a Lightning checkpoint written by the reference (torch.optim.Adam) loads into FusedAdam and back."""
    rs = np.random.RandomState(2)
    mk = lambda: [torch.nn.Parameter(torch.as_tensor(rs.randn(n).astype(np.float32), device=DEV)) for n in (1000, 24)]
    A, B = mk(), mk()
    for a, b in zip(A, B):
        b.data.copy_(a.data)
    ta = torch.optim.Adam([{"params": [A[0]]}, {"params": [A[1]], "lr": 1e-5}], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    fb = FusedAdam([{"params": [B[0]]}, {"params": [B[1]], "lr": 1e-5}], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    for k in range(3):
        for a in A:
            a.grad = torch.as_tensor(rs.randn(*a.shape).astype(np.float32), device=DEV)
        ta.step()
    fb.load_state_dict(ta.state_dict())
    assert float(fb.state[B[0]]["step"]) == 3.0 and fb.state[B[0]]["step"].is_cuda
    for a, b in zip(A, B):
        b.data.copy_(a.data)
        g = torch.as_tensor(rs.randn(*a.shape).astype(np.float32), device=DEV)
        a.grad, b.grad = g.clone(), g.clone()
    ta.step()
    fb.step()
    for a, b in zip(A, B):   # (torch's CUDA kernels round sqrt / division differently in the last bit)
        assert (a.detach() - b.detach()).abs().max() < 1e-2 * 3e-6 and torch.equal(ta.state[a]["exp_avg"], fb.state[b]["exp_avg"]) or (ta.state[a]["exp_avg"] - fb.state[b]["exp_avg"]).abs().max() < 1e-7
    back = torch.optim.Adam([{"params": [A[0]]}, {"params": [A[1]], "lr": 1e-5}], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    back.load_state_dict(fb.state_dict())
    assert float(back.state[A[0]]["step"]) == 4.0
