#!/usr/bin/env python
"""The pin for SURVEY.md rows a10 / a11 (tcnn HashGrid + FullyFusedMLP): ONE command on any box where `import tinycudann`
works (tiny-cuda-nn v1.6, the version /root/reference/install.sh:6 installs; needs an NVIDIA GPU -- this repo's MI355X
boxes cannot run it, which is why the rows say "parity unpinned"):

    python tools/make_tcnn_golden.py                     # -> tests/golden/tcnn_golden.npz  (~1 MB; commit it)

It builds the reference's two modules with the reference's configs (instant_avatar/models/networks/ngp.py:27-58), replaces
their parameters by a seeded vector this script can regenerate anywhere (values exactly representable in fp16), evaluates
4 096 points of the unit cube -- random ones, the 8 corners, points on faces and edges, points on and next to the grid lines
of level 3 (whose resolution, 54 or 55, is the one open question of the layout: it hangs on the last bit of that build's
exp2f) -- exactly the way NeRFNGPNet.forward does (ngp.py:78-81), runs one backward, and stores

    n_enc, n_col      params.numel() of the two modules (decides the level-3 layout)
    points            [4096,3] fp32
    feat              [4096,32] fp16   tcnn.Encoding(HashGrid) on the same grid parameters (a10, expected BIT-exact)
    enc_out           [4096,16] fp16   NetworkWithInputEncoding output: sigma = [:,0]     (a11)
    col_out           [4096,3]  fp16   color_net(enc_out[:,1:])                           (a11)
    g_enc_mlp, g_col  gradients of the MLP weights, g_grid_idx / g_grid_val the non-zero grid gradients, for the loss
                      sum(sigma[:256] * w_sigma) + sum(col_out[:256] * w_col) with the stored weights (sigma = enc_out[:,0]:
                      the two outputs NeRFNGPNet.forward returns, ngp.py:79-83)
    meta              tcnn / torch versions, GPU name, seed

The consumers: tests/test_cpu_oracle.py::test_tcnn_golden (the CPU oracle's restatement against the file) and
tests/test_gpu_tcnn_golden.py (the HIP kernels against the file).  Both XFAIL with "parity unpinned" while the file is absent.
`--self-made PATH` writes a file of the same shape from the CPU oracle instead of tcnn: it exercises the consumers (the repo's
own tests do that) and is marked so that no consumer can mistake it for the pin.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_OUT = os.path.join(ROOT, "tests", "golden", "tcnn_golden.npz")
SEED = 20240807
N_POINTS = 4096
N_BWD = 256
SIG_W1, SIG_W2 = 64 * 32, 16 * 64           # encoder MLP: 32 -> 64 -> 16 (ngp.py:38-44), matrices [out x in]
COL_SIZES = (64 * 16, 64 * 64, 16 * 64)     # colour MLP: 16 (15 + pad) -> 64 -> 64 -> 16 (3 used) (ngp.py:47-58)

ENCODING = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16,
            "per_level_scale": 1.5}
NET_SIGMA = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 1}
NET_COLOR = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "Sigmoid", "n_neurons": 64, "n_hidden_layers": 2}


def _fp16_exact(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def golden_params(n_enc, n_col, seed=SEED):
    """The parameter vectors of the two modules, as a function of their sizes alone: MLP weights Xavier-uniform, grid
    U(-0.5, 0.5) (tcnn's own 1e-4 initialisation gives features that round to nothing), every value an fp16 number."""
    rng = np.random.RandomState(seed)
    enc = np.empty(n_enc, np.float32)
    o = 0
    for out_f, in_f in ((64, 32), (16, 64)):
        a = (6.0 / (out_f + in_f)) ** 0.5
        enc[o:o + out_f * in_f] = rng.uniform(-a, a, out_f * in_f)
        o += out_f * in_f
    enc[o:] = rng.uniform(-0.5, 0.5, n_enc - o)
    col = np.empty(n_col, np.float32)
    o = 0
    for out_f, in_f in ((64, 16), (64, 64), (16, 64)):
        a = (6.0 / (out_f + in_f)) ** 0.5
        col[o:o + out_f * in_f] = rng.uniform(-a, a, out_f * in_f)
        o += out_f * in_f
    assert o == n_col, "colour MLP with %d parameters (expected %d)" % (n_col, o)
    return _fp16_exact(enc), _fp16_exact(col)


def golden_points(n=N_POINTS, seed=SEED):
    """[n,3] fp32 in [0,1]: corners, faces, edges, the grid lines of level 3 (scale 53: k / 53 and one ulp either side, the
    upper rim 1 - 2^-24 ... 1 where resolution 54 and 55 index differently), then uniform random points."""
    rng = np.random.RandomState(seed + 1)
    pts = [np.array([[i, j, k] for i in (0, 1) for j in (0, 1) for k in (0, 1)], np.float32)]          # 8 corners
    f = rng.uniform(0, 1, (192, 3)).astype(np.float32)                                                  # faces: one coordinate pinned
    f[np.arange(192), np.arange(192) % 3] = (np.arange(192) // 3 % 2).astype(np.float32)
    pts.append(f)
    e = rng.uniform(0, 1, (96, 3)).astype(np.float32)                                                   # edges: two pinned
    for r in range(96):
        a, b = r % 3, (r + 1) % 3
        e[r, a], e[r, b] = float(r // 3 % 2), float(r // 6 % 2)
    pts.append(e)
    k = rng.randint(0, 54, (600, 3)).astype(np.float32)
    line = (k / np.float32(53.0)).astype(np.float32)                                                    # level 3: pos = x * 53 + 0.5
    line = np.where(rng.rand(600, 3) < 0.33, np.nextafter(line, np.float32(2)), np.where(rng.rand(600, 3) < 0.5, np.nextafter(line, np.float32(-1)), line))
    pts.append(np.clip(line, 0, 1).astype(np.float32))
    rim = rng.uniform(0, 1, (200, 3)).astype(np.float32)
    rim[np.arange(200), np.arange(200) % 3] = np.float32(1) - np.float32(2.0) ** -np.float32(rng.randint(10, 25, 200))
    pts.append(rim)
    have = sum(len(p) for p in pts)
    pts.append(rng.uniform(0, 1, (n - have, 3)).astype(np.float32))
    out = np.concatenate(pts).astype(np.float32)
    assert out.shape == (n, 3) and out.min() >= 0 and out.max() <= 1
    return out


def loss_weights(seed=SEED):
    rng = np.random.RandomState(seed + 2)
    return rng.uniform(-1, 1, N_BWD).astype(np.float32), rng.uniform(-1, 1, (N_BWD, 3)).astype(np.float32)


def run_tcnn(points):
    import torch
    import tinycudann as tcnn
    dev = "cuda"
    enc = tcnn.NetworkWithInputEncoding(n_input_dims=3, n_output_dims=16, encoding_config=ENCODING, network_config=NET_SIGMA)
    col = tcnn.Network(n_input_dims=15, n_output_dims=3, network_config=NET_COLOR)
    grid = tcnn.Encoding(n_input_dims=3, encoding_config=ENCODING)
    n_enc, n_col = enc.params.numel(), col.params.numel()
    assert grid.params.numel() == n_enc - SIG_W1 - SIG_W2, "NetworkWithInputEncoding.params is not [W1 | W2 | grid]: %d vs %d" % (grid.params.numel(), n_enc)
    p_enc, p_col = golden_params(n_enc, n_col)
    with torch.no_grad():
        enc.params.copy_(torch.from_numpy(p_enc).to(enc.params))
        col.params.copy_(torch.from_numpy(p_col).to(col.params))
        grid.params.copy_(torch.from_numpy(p_enc[SIG_W1 + SIG_W2:]).to(grid.params))
    x = torch.from_numpy(points).to(dev)
    with torch.no_grad():
        feat = grid(x)
    x_enc = enc(x)                        # ngp.py:78
    c = col(x_enc[..., 1:])               # ngp.py:81
    w_sigma, w_col = (torch.from_numpy(w).to(dev) for w in loss_weights())
    loss = (x_enc[:N_BWD, 0].float() * w_sigma).sum() + (c[:N_BWD].float() * w_col).sum()
    loss.backward()
    g_enc = enc.params.grad.float().cpu().numpy()
    g_grid = g_enc[SIG_W1 + SIG_W2:]
    nz = np.flatnonzero(g_grid)
    meta = "tinycudann %s, torch %s, %s, seed %d" % (getattr(tcnn, "__version__", "?"), torch.__version__, torch.cuda.get_device_name(0), SEED)
    return dict(n_enc=n_enc, n_col=n_col, points=points, feat=feat.half().cpu().numpy(), enc_out=x_enc.detach().half().cpu().numpy(),
                col_out=c.detach().half().cpu().numpy(), g_enc_mlp=g_enc[:SIG_W1 + SIG_W2].astype(np.float32),
                g_col=col.params.grad.float().cpu().numpy().astype(np.float32), g_grid_idx=nz.astype(np.int64), g_grid_val=g_grid[nz].astype(np.float32),
                w_sigma=w_sigma.cpu().numpy(), w_col=w_col.cpu().numpy(), meta=np.array(meta), source=np.array("tinycudann"))


def field_dict(p_enc, p_col, n_levels=16):
    """the flat tcnn vectors as the named arrays oracle.make_field / NeRFNGPNet.load_field_dict take; the unit cube is the box"""
    f16 = lambda a: np.ascontiguousarray(np.asarray(a, np.float32).astype(np.float16))
    return dict(center=np.full(3, 0.5, np.float32), scale=np.ones(3, np.float32), n_levels=n_levels, log2_T=19,
                sig_w1=f16(p_enc[:SIG_W1]), sig_w2=f16(p_enc[SIG_W1:SIG_W1 + SIG_W2]), table=f16(p_enc[SIG_W1 + SIG_W2:]).reshape(-1, 2),
                col_w1=f16(p_col[:COL_SIZES[0]]), col_w2=f16(p_col[COL_SIZES[0]:COL_SIZES[0] + COL_SIZES[1]]), col_w3=f16(p_col[COL_SIZES[0] + COL_SIZES[1]:]))


def level3_res_of(n_enc):
    """54 or 55, from the size of encoder.params (the grid's level table is fixed but for that one resolution)"""
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    for r3 in (54, 55):
        hd = orc.hash_desc(16, 19, level3_res=r3)
        if SIG_W1 + SIG_W2 + 2 * int(hd.offset[16]) == int(n_enc):
            return r3
    raise ValueError("encoder.params with %d elements fits neither level-3 layout" % n_enc)


def run_oracle(points, level3_res=None):
    """the same record from the CPU oracle (oracle/ia_oracle.c): NOT a pin -- it lets the consumers be exercised without tcnn"""
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    r3 = int(level3_res or os.environ.get("IA_TCNN_LEVEL3_RES", "54"))
    hd = orc.hash_desc(16, 19, level3_res=r3)
    n_enc, n_col = SIG_W1 + SIG_W2 + 2 * int(hd.offset[16]), sum(COL_SIZES)
    p_enc, p_col = golden_params(n_enc, n_col)
    old = os.environ.get("IA_TCNN_LEVEL3_RES")
    os.environ["IA_TCNN_LEVEL3_RES"] = str(r3)
    try:
        field, keep = orc.make_field(field_dict(p_enc, p_col))
        feat = orc.hashgrid(field, points)
        enc_out = orc.tcnn_encoder(field, points)
        col_out = orc.tcnn_color(field, enc_out[:, 1:])
    finally:
        if old is None:
            os.environ.pop("IA_TCNN_LEVEL3_RES", None)
        else:
            os.environ["IA_TCNN_LEVEL3_RES"] = old
    w_sigma, w_col = loss_weights()
    z = np.zeros(0, np.float32)
    return dict(n_enc=n_enc, n_col=n_col, points=points, feat=np.asarray(feat, np.float16), enc_out=enc_out.astype(np.float16),
                col_out=col_out.astype(np.float16), g_enc_mlp=z, g_col=z, g_grid_idx=np.zeros(0, np.int64), g_grid_val=z, w_sigma=w_sigma, w_col=w_col,
                meta=np.array("self-made by the CPU oracle (level-3 resolution %d): exercises the consumers, pins nothing" % r3),
                source=np.array("oracle"))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--out", default=DEFAULT_OUT)
    ap.add_argument("--self-made", metavar="PATH", help="write a consumer-exercise file from the CPU oracle instead (never to the default path)")
    args = ap.parse_args(argv)
    pts = golden_points()
    if args.self_made:
        if os.path.abspath(args.self_made) == os.path.abspath(DEFAULT_OUT):
            ap.error("a self-made file must not take the place of the pin")
        np.savez_compressed(args.self_made, **run_oracle(pts))
        print("wrote", args.self_made, "(self-made: not a pin)")
        return 0
    try:
        import tinycudann  # noqa: F401
    except Exception as e:
        raise SystemExit("make_tcnn_golden: `import tinycudann` failed (%s).  Run this on a box with tiny-cuda-nn v1.6 "
                         "(pip install git+https://github.com/NVlabs/tiny-cuda-nn/@v1.6#subdirectory=bindings/torch) and an NVIDIA GPU." % (e,))
    rec = run_tcnn(pts)
    np.savez_compressed(args.out, **rec)
    print("wrote %s: encoder.params %d (level-3 resolution %d), color_net.params %d, %d non-zero grid gradients -- commit it; "
          "`pytest tests/test_cpu_oracle.py -k tcnn_golden` and `pytest -m gpu tests/test_gpu_tcnn_golden.py` now pin rows a10 / a11"
          % (args.out, rec["n_enc"], level3_res_of(rec["n_enc"]), rec["n_col"], len(rec["g_grid_idx"])))
    return 0


if __name__ == "__main__":
    sys.exit(main())
