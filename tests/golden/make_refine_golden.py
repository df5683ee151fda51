"""Generates tests/golden/refine_golden.npz: BASELINE config 4 (confs/SNARF_NGP_refine.yaml -- SMPL parameters optimised
together with the field through the SNARF deformer) as the REFERENCE's `DNeRFModel.training_step` EXECUTING on the CPU
(tests/golden/ref_cpu_harness.py with the differentiable tcnn stand-ins):

    DNeRF.py:112-161   SMPLParamEmbedding rows -> batch, near / far from the refined translation, prepare_deformer under
                       autograd (lbs.py), update_density_grid (step 0), forward -> Raymarcher.render_train -> deform_train ->
                       ForwardDeformer.forward (implicit differentiation, deformer_torch.py:50-67), NGPLoss, no sigma noise
                       and no density regulariser (is_refine), GradScaler(1024) / Adam with the three parameter groups
                       (lr 1e-2, 1e-2, optimize_SMPL.lr = 1e-5)

for N_STEPS consecutive steps on N_STEPS frames.  Recorded per step: the five loss values, d loss / d tfs, the gradients
of the four SMPL tables, of the five MLP weight matrices and (sampled + norm) of the hash table, the rendered rgb / alpha;
at the end the SMPL tables.  tests/test_gpu_refine.py replays the same steps through the HIP product path.

Run from the repo root:  python tests/golden/make_refine_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
# IA_GOLDEN_BLEND=1: the same recipe on the blend-shape body (synthetic.make_body(blendshapes=True): non-zero shapedirs / posedirs,
# dense J_regressor) with the shape coefficients synthetic.BLEND_BETAS -> *_blend.npz
BLEND = os.environ.get("IA_GOLDEN_BLEND", "0") == "1"
OUT = os.path.join(HERE, "refine_golden%s%s.npz" % ("_blend" if BLEND else "", "_l55" if os.environ.get("IA_TCNN_LEVEL3_RES", "54") == "55" else ""))
RES, N_RAYS, N_STEPS, N_FRAMES = 32, 768, 3, 4
SEED_DRAWS, SEED_SEL, SEED_PERTURB = 404, 17, 23
POSE_NOISE, TRANSL_NOISE = 0.03, 0.01


def scenario():
    """What both sides start from: the true pose track (targets are rendered at it), the perturbed SMPL tables (what is
    being optimised), the ray selection per step."""
    from instantavatar_amd import synthetic as syn
    poses, tr = syn.procedural_pose_track(8)
    rs = np.random.RandomState(SEED_PERTURB)
    tables = dict(betas=(syn.BLEND_BETAS.reshape(1, 10).copy() if BLEND else np.zeros((1, 10), np.float32)),
                  global_orient=(poses[:N_FRAMES, :3] + POSE_NOISE * rs.randn(N_FRAMES, 3)).astype(np.float32),
                  body_pose=(poses[:N_FRAMES, 3:] + POSE_NOISE * rs.randn(N_FRAMES, 69)).astype(np.float32),
                  transl=(tr[:N_FRAMES] + TRANSL_NOISE * rs.randn(N_FRAMES, 3)).astype(np.float32))
    sel = np.stack([np.random.RandomState(SEED_SEL + k).permutation(RES * RES)[:N_RAYS] for k in range(N_STEPS)])
    return poses, tr, tables, sel


def main():
    import ref_cpu_harness as H
    from instantavatar_amd import synthetic as syn
    from oracle import oracle
    body = syn.make_body(blendshapes=BLEND)
    betas = syn.BLEND_BETAS if BLEND else np.zeros(10, np.float32)
    init = oracle.deformer_initialize(body, betas, syn.cano_pose("A_pose"), resolution=32, n_smooth=30)
    fp = syn.make_field(init["cano_joints"], init["bbox"])
    holder = {"fp": fp}
    holder["field"], holder["keep"] = oracle.make_field(fp)
    lp = types.ModuleType("third_parties.lpips")          # NGPLoss constructs it (loss.py:11); w_lpips is 0 in the refine config
    lp.LPIPS = lambda **kw: torch.nn.Identity()
    sys.modules["third_parties.lpips"] = lp
    R = H.install(oracle, holder, differentiable=True)
    import instant_avatar.utils.loss as ref_loss
    from instant_avatar.models.structures.body_model_param import SMPLParamEmbedding
    model = H.build_reference_model(R, body, fp, resolution=32)
    poses, tr, tables, sel = scenario()
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32))
    # ---- DNeRFModel.__init__ (DNeRF.py:18-30) for the refine configuration ----
    model.opt = H.Opt(optimize_SMPL=H.Opt(enable=True, is_refine=True, lr=1e-5), optimizer=H.Opt(lr=1e-2, betas=(0.9, 0.99), eps=1e-15),
                      scheduler=H.Opt(max_epochs=20))
    # (copies: torch.as_tensor shares memory with the numpy arrays, and Adam updates the embedding weights in place)
    model.SMPL_param = SMPLParamEmbedding(**{k: t(v.copy()) for k, v in tables.items()})
    model.loss_fn = ref_loss.NGPLoss(H.Opt(w_rgb=1.0, w_alpha=0.1, w_reg=0.1))
    model.datamodule = types.SimpleNamespace(trainset=types.SimpleNamespace(smpl_params=tables))
    model.automatic_optimization = False
    model.precision = 32
    model.log = lambda *a, **k: None
    (optim,), _ = R.dnerf.DNeRFModel.configure_optimizers(model)       # also creates model.scaler (GradScaler, disabled without CUDA)
    model.optimizers = lambda *a: optim

    class _Raise:
        def warning(self, e):
            raise e                                                     # DNeRF.py:160 swallows exceptions: not in a golden
    R.dnerf.logger = _Raise()
    # tfs is not a leaf: keep its gradient
    prep = model.deformer.prepare_deformer

    def prepare_and_retain(params):
        prep(params)
        model.deformer.tfs.retain_grad()
    model.deformer.prepare_deformer = prepare_and_retain
    # record what training_step computes but does not return
    fwd, loss_call = R.dnerf.DNeRFModel.forward, model.loss_fn.forward
    rec = {}

    def forward_rec(self, batch, eval_mode=False):
        rec["pred"] = fwd(self, batch, eval_mode)
        return rec["pred"]

    def loss_rec(pred, tgt):
        rec["losses"] = loss_call(pred, tgt)
        return rec["losses"]
    R.dnerf.DNeRFModel.forward = forward_rec
    model.loss_fn.forward = loss_rec

    # ---- targets: the oracle's test render at the TRUE pose of each frame ----
    ro, rd = syn.make_camera_rays(RES)
    out = dict(res=np.int32(RES), n_rays=np.int32(N_RAYS), n_steps=np.int32(N_STEPS), n_frames=np.int32(N_FRAMES), sel=sel,
               seeds=np.array([SEED_DRAWS, SEED_SEL, SEED_PERTURB]), **{"table_" + k: v for k, v in tables.items()})
    model.train()
    model.global_step = 0
    n1 = 64 * 32
    with H.SeededDraws() as draws:
        for k in range(N_STEPS):
            f = k % N_FRAMES
            world = oracle.make_world(body, init, fp, betas, poses[f, 3:], poses[f, :3], tr[f], syn.INIT_BONES)
            jit = np.random.RandomState(SEED_DRAWS + 100 + k).rand(2, 64 ** 3, 3).astype(np.float32)
            tgt = oracle.render_image_fast(world, ro, rd, jit)
            s = sel[k]
            dist = float(np.sqrt((tr[f] ** 2).sum()))
            batch = {"rays_o": t(ro[s])[None], "rays_d": t(rd[s])[None], "near": torch.full((1, N_RAYS), dist - 1), "far": torch.full((1, N_RAYS), dist + 1),
                     "betas": t(betas)[None], "rgb": t(tgt["rgb"][s])[None], "alpha": t(tgt["alpha"][s])[None],
                     "bg_color": torch.ones(1, N_RAYS, 3), "idx": torch.tensor([f])}
            out["tgt_rgb_%d" % k], out["tgt_alpha_%d" % k] = tgt["rgb"][s], tgt["alpha"][s]
            draws.seed(SEED_DRAWS + k)
            R.dnerf.DNeRFModel.training_step(model, batch)
            model.global_step += 1                                       # Lightning's loop
            L = rec["losses"]
            out["loss_%d" % k] = np.array([float(L[n].detach()) for n in ("loss", "mse_loss", "loss_alpha_coarse", "reg_alpha", "reg_density")], np.float64)
            out["rgb_%d" % k] = rec["pred"]["rgb_coarse"].detach().numpy()[0]
            out["alpha_%d" % k] = rec["pred"]["alpha_coarse"].detach().numpy()[0]
            out["d_tfs_%d" % k] = model.deformer.tfs.grad.numpy()[0].copy()
            out["tfs_%d" % k] = model.deformer.tfs.detach().numpy()[0].copy()
            for name in ("betas", "global_orient", "transl", "body_pose"):
                g = getattr(model.SMPL_param, name).weight.grad
                # (betas: None -- with the SNARF deformer the batch keeps the data set's betas, DNeRF.py:121-123)
                out["g_%s_%d" % (name, k)] = g.numpy().copy() if g is not None else np.zeros((0,), np.float32)
            ge, gc = model.net_coarse.encoder.params.grad.numpy(), model.net_coarse.color_net.params.grad.numpy()
            out["g_mlp_sigma_%d" % k] = ge[:n1 + 1024].copy()
            out["g_mlp_color_%d" % k] = gc.copy()
            gt = ge[n1 + 1024:]
            nz = np.flatnonzero(gt)
            out["g_table_norm_%d" % k] = np.float64(np.sqrt((gt.astype(np.float64) ** 2).sum()))
            out["g_table_nnz_%d" % k] = np.int64(len(nz))
            pick = nz[:: max(len(nz) // 4096, 1)][:4096]
            out["g_table_idx_%d" % k], out["g_table_val_%d" % k] = pick.astype(np.int64), gt[pick].copy()
            print("step %d frame %d: loss %.6f mse %.6f | |d_tfs| %.4e | |g body_pose| %.4e |g orient| %.4e |g transl| %.4e | table nnz %d" % (
                k, f, out["loss_%d" % k][0], out["loss_%d" % k][1], np.abs(out["d_tfs_%d" % k]).sum(),
                np.abs(out["g_body_pose_%d" % k]).sum(), np.abs(out["g_global_orient_%d" % k]).sum(), np.abs(out["g_transl_%d" % k]).sum(), len(nz)))
    for name in ("betas", "global_orient", "transl", "body_pose"):
        out["final_" + name] = getattr(model.SMPL_param, name).weight.detach().numpy().copy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
