"""Rows a10 / a11 on the device: the HIP encoder and the MFMA field kernels against outputs of tiny-cuda-nn v1.6 itself
(tests/golden/tcnn_golden.npz, tools/make_tcnn_golden.py).  XFAIL "parity unpinned" while that file is absent; the consumer is
exercised on a record written by the CPU oracle (not a pin) so that it is known to run the day the real file arrives."""
import numpy as np
import pytest
import torch

import tcnn_golden as TG

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _net(g):
    from instantavatar_amd.models.networks.ngp import NeRFNGPNet
    r3, p_enc, p_col = TG.params_of(g)
    net = NeRFNGPNet(dict(center=[0.5, 0.5, 0.5], scale=[1, 1, 1]), level3_res=r3).to(DEV)
    assert net.encoder.params.numel() == int(g["n_enc"]) and net.color_net.params.numel() == int(g["n_col"])      # the layout question, answered by numel
    net.load_tcnn_params(torch.from_numpy(p_enc), torch.from_numpy(p_col))
    return net, r3


def check_device(g):
    net, r3 = _net(g)
    x = torch.as_tensor(np.ascontiguousarray(g["points"], np.float32), device=DEV)
    res = {"level3_res": r3}
    feat = net.encode(x).cpu().numpy()
    ref_f = np.asarray(g["feat"], np.float16)
    res["feat_mismatch"] = int((feat.view(np.uint16) != ref_f.view(np.uint16)).sum())
    with torch.no_grad():
        rgb, sigma = net(x)
    ref_s, ref_c = g["enc_out"].astype(np.float32)[:, 0], g["col_out"].astype(np.float32)
    res["sigma_max_rel"] = float((np.abs(sigma.cpu().numpy() - ref_s) / np.maximum(1.0, np.abs(ref_s))).max())
    res["rgb_max_abs"] = float(np.abs(rgb.cpu().numpy() - ref_c).max())
    if len(g["g_col"]):
        # one backward of the loss the record was made with (tools/make_tcnn_golden.py)
        n = len(g["w_sigma"])
        net.train()
        for p in net.parameters():
            p.grad = None
        with torch.enable_grad():
            rgb, sigma = net(x)
            loss = (sigma[:n] * torch.as_tensor(g["w_sigma"], device=DEV)).sum() + (rgb[:n] * torch.as_tensor(g["w_col"], device=DEV)).sum()
            loss.backward()
        ge, gc = net.encoder.params.grad.cpu().numpy(), net.color_net.params.grad.cpu().numpy()
        n_mlp = len(g["g_enc_mlp"])
        cos = lambda a, b: float(np.dot(a, b) / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-30))
        res["g_enc_mlp_cos"], res["g_col_cos"] = cos(ge[:n_mlp], g["g_enc_mlp"]), cos(gc, g["g_col"])
        grid = ge[n_mlp:]
        idx = g["g_grid_idx"]
        res["g_grid_cos"] = cos(grid[idx], g["g_grid_val"])
        mask = np.ones(len(grid), bool)
        mask[idx] = False
        res["g_grid_outside_support"] = float(np.abs(grid[mask]).max()) if mask.any() else 0.0
    return res


def assert_device(res):
    assert res["feat_mismatch"] == 0, res                         # a10: bit-exact fp16 features
    assert res["sigma_max_rel"] < 4e-3 and res["rgb_max_abs"] < 2e-3, res      # a11: two half roundings (fp32 accumulation here, half in tcnn)
    if "g_col_cos" in res:
        assert res["g_enc_mlp_cos"] > 0.999 and res["g_col_cos"] > 0.999 and res["g_grid_cos"] > 0.999, res
        assert res["g_grid_outside_support"] == 0.0, res          # no gradient lands on a table entry tcnn did not touch


def test_tcnn_golden_on_the_device():
    g = TG.load()
    if g is None or not g["is_pin"]:
        pytest.xfail(TG.UNPINNED)
    res = check_device(g)
    print("tcnn golden on the device (%s): %s" % (g["meta"], res))
    assert_device(res)


@pytest.mark.parametrize("r3", [54, 55])
def test_tcnn_golden_device_consumer_on_a_self_made_file(tmp_path, r3):
    """the device-side consumer on a record written by the CPU oracle: HIP == oracle on the golden's points (corners, faces,
    the level-3 grid lines) in both layouts; the forward comparisons all run.  Not a pin."""
    t = TG.tool()
    path = str(tmp_path / "self.npz")
    np.savez_compressed(path, **t.run_oracle(t.golden_points(), level3_res=r3))
    g = TG.load(path)
    assert not g["is_pin"]
    res = check_device(g)
    print("self-made (level-3 %d): %s" % (r3, res))
    assert res["level3_res"] == r3
    assert_device(res)
