"""Generates tests/golden/pipeline_golden.npz: outputs of the REFERENCE's Python pipeline executing on the CPU
(tests/golden/ref_cpu_harness.py: instant_avatar.* imported from /root/reference, native extensions / tcnn replaced by
adapters around the oracle's C functions):

  (A) DNeRFModel.render_image_fast (DNeRF.py:72-97): prepare_deformer, DensityGrid.initialize (5 jittered probe sets,
      max-pool, threshold, largest component), transform_rays_w2s, Raymarcher.render_test -- a 32 x 32 frame;
  (B) DNeRFModel.update_density_grid (DNeRF.py:99-110) -> DensityGrid.update (EMA, post-processing, step < 500 switch),
      two consecutive updates, through deform_train with the training branch of ForwardDeformer.forward;
  (C) DNeRFModel.forward in training mode -> Raymarcher.render_train with jitter and sigma noise on 192 rays;
  (D) the skinning-weight voxels the reference's switch_to_explicit builds (KNN + 30 smoothing passes in torch).

tests/test_cpu_oracle.py::test_oracle_pipeline_matches_reference_python_golden re-creates the random draws from the
seeds below and compares the oracle's own pipeline functions.   Run from the repo root:  python tests/golden/make_pipeline_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
# the hash-grid layout switch (tcnn level-3 resolution 54 / 55) changes the field: one golden per layout
# IA_GOLDEN_BLEND=1: the same recipe on the blend-shape body (synthetic.make_body(blendshapes=True): non-zero shapedirs / posedirs,
# dense J_regressor) with the shape coefficients synthetic.BLEND_BETAS -> *_blend.npz
BLEND = os.environ.get("IA_GOLDEN_BLEND", "0") == "1"
OUT = os.path.join(HERE, "pipeline_golden%s%s.npz" % ("_blend" if BLEND else "", "_l55" if os.environ.get("IA_TCNN_LEVEL3_RES", "54") == "55" else ""))
SEED_INIT, SEED_UPD, SEED_TRAIN = 101, 202, 303
RES, FRAME, N_TRAIN = 32, 1, 192


def main():
    import ref_cpu_harness as H
    from instantavatar_amd import synthetic as syn
    from oracle import oracle
    body = syn.make_body(blendshapes=BLEND)
    betas = syn.BLEND_BETAS if BLEND else np.zeros(10, np.float32)
    init = oracle.deformer_initialize(body, betas, syn.cano_pose("A_pose"), resolution=32, n_smooth=30)
    fp = syn.make_field(init["cano_joints"], init["bbox"])
    field, keep = oracle.make_field(fp)
    holder = {"field": field}
    R = H.install(oracle, holder)
    model = H.build_reference_model(R, body, fp, resolution=32)
    poses, tr = syn.procedural_pose_track(8)
    ro, rd = syn.make_camera_rays(RES)
    dist = float(np.sqrt((tr[FRAME] ** 2).sum()))
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32))
    batch = {"rays_o": t(ro)[None], "rays_d": t(rd)[None], "near": torch.full((1, RES * RES), dist - 1), "far": torch.full((1, RES * RES), dist + 1),
             "betas": t(betas)[None], "body_pose": t(poses[FRAME, 3:])[None], "global_orient": t(poses[FRAME, :3])[None],
             "transl": t(tr[FRAME])[None]}
    out = {}
    with H.SeededDraws() as draws:
        # ---- (A) ----
        model.eval()
        draws.seed(SEED_INIT)
        rgb, depth, alpha, counter = R.dnerf.DNeRFModel.render_image_fast(model, dict(batch), (RES, RES))
        g = model.renderer.density_grid_test
        out.update(A_rgb=rgb.numpy()[0], A_depth=depth.numpy()[0], A_alpha=alpha.numpy()[0], A_counter=counter.numpy()[0],
                   A_occ=np.packbits(g.density_field.numpy().astype(np.uint8)), A_aabb=torch.stack(list(g.aabb)).numpy(),
                   tfs=model.deformer.tfs.numpy()[0], w2s=model.deformer.w2s.numpy()[0], bbox=model.deformer.bbox.numpy())
        print("(A) alpha coverage %.3f, occupied cells %d, samples/ray %.2f" % (float((alpha > 0.5).float().mean()), int(g.density_field.sum()), float(counter.mean())))
        # ---- (A2): the same frame and occupancy grid with MAX_BATCH_SIZE = 4096, so that the wave-front loop of
        # Raymarcher.render_test runs many iterations with a changing N_step (4, 5, 6, ... as rays retire) ----
        model.renderer.MAX_BATCH_SIZE = 4096
        d2 = R.dnerf.DNeRFModel.forward(model, dict(batch))
        model.renderer.MAX_BATCH_SIZE = 291600
        out.update(A2_rgb=d2["rgb_coarse"].numpy()[0], A2_depth=d2["depth_coarse"].numpy()[0], A2_alpha=d2["alpha_coarse"].numpy()[0],
                   A2_counter=d2["counter_coarse"].numpy()[0])
        print("(A2) max |rgb - rgb(A)| %.3e, samples/ray %.2f" % (float((d2["rgb_coarse"][0] - rgb[0].reshape(-1, 3)).abs().max()), float(d2["counter_coarse"].mean())))
        # ---- (D) ----
        lbs_ref = model.deformer.deformer.lbs_voxel_final.numpy()[0]
        out["D_lbs_max_abs_diff_to_oracle"] = np.float32(np.abs(lbs_ref - init["lbs_voxel"]).max())
        out["D_lbs_sample"] = lbs_ref.reshape(24, -1)[:, ::97].copy()
        print("(D) skinning-weight voxels, reference torch build vs oracle C build: max abs diff %.3e" % float(out["D_lbs_max_abs_diff_to_oracle"]))
        # ---- (B) ----
        model.train()
        for k, step in enumerate((0, 500)):
            model.global_step = step
            draws.seed(SEED_UPD + k)
            reg = R.dnerf.DNeRFModel.update_density_grid(model)
            gt = model.renderer.density_grid_train
            out["B%d_reg" % k] = np.float32(float(reg))
            out["B%d_field" % k] = np.packbits(gt.density_field.numpy().astype(np.uint8))
            out["B%d_cached_sample" % k] = gt.density_cached.numpy().reshape(-1)[::61].copy()
            out["B%d_cached_sum" % k] = np.float64(gt.density_cached.double().sum())
            print("(B) step %d: reg %.6e, field %d cells, cached sum %.4f" % (step, float(reg), int(gt.density_field.sum()), float(out["B%d_cached_sum" % k])))
        # ---- (C) ----
        model.global_step = 20
        sel = np.random.RandomState(7).permutation(RES * RES)[:N_TRAIN]
        tb = dict(batch)
        for k in ("rays_o", "rays_d"):
            tb[k] = batch[k][:, sel]
        for k in ("near", "far"):
            tb[k] = batch[k][:, sel]
        tb["bg_color"] = t(np.random.RandomState(8).rand(1, N_TRAIN, 3))
        draws.seed(SEED_TRAIN)
        d = R.dnerf.DNeRFModel.forward(model, tb)
        out.update(C_sel=sel, C_bg=tb["bg_color"].numpy()[0], C_rgb=d["rgb_coarse"].detach().numpy()[0], C_alpha=d["alpha_coarse"].detach().numpy()[0],
                   C_depth=d["depth_coarse"].detach().numpy()[0], C_weights=d["weight_coarse"].detach().numpy()[0])
        print("(C) train render: alpha mean %.4f" % float(d["alpha_coarse"].mean()))
    # (F) DNeRFModel.configure_optimizers (DNeRF.py:32-59): parameter groups and the LambdaLR schedule
    model.opt["optimizer"] = H.Opt(lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    model.opt["scheduler"] = H.Opt(max_epochs=30)
    (optim,), (sched,) = R.dnerf.DNeRFModel.configure_optimizers(model)
    names = {id(p): n for n, p in model.named_parameters()}
    out["F_groups"] = np.array(["%s|%g|%s|%g" % (",".join(names[id(p)] for p in g["params"]), g["lr"], g["betas"], g["eps"]) for g in optim.param_groups])
    fac = []
    for epoch in range(31):
        fac.append(optim.param_groups[0]["lr"])
        optim.step()
        sched.step()
    out["F_lr_per_epoch"] = np.array(fac, np.float64)
    print("(F) groups", list(out["F_groups"]), "lr[0,1,15,29,30]", [fac[i] for i in (0, 1, 15, 29, 30)])
    # (E) the checkpoint surface: what the reference's modules register (tcnn's two flat vectors are 1-element stand-ins here)
    sd = model.state_dict()
    out["E_state_dict"] = np.array(["%s|%s|%s" % (k, "x".join(str(d) for d in v.shape), str(v.dtype)) for k, v in sd.items()])
    np.savez_compressed(OUT, seeds=np.array([SEED_INIT, SEED_UPD, SEED_TRAIN]), res=np.int32(RES), frame=np.int32(FRAME), **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
