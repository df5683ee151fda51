#!/usr/bin/env python3
"""bench.py -- novel-pose rendering throughput of the InstantAvatar hot path on MI355X.

A "step" is one frame of DNeRFModel.render_image_fast (models/DNeRF.py:72-97) at
512x512: SMPL joint chain -> skinning-voxel precompute -> per-frame occupancy
build (5 x 64^3 probes through the deformer + field) -> occupancy-grid ray march
with Fast-SNARF root finding, hash-grid + MLP field and alpha compositing.
Inputs (camera rays, SMPL poses, weights) are resident in HBM before the timed
region.  Synthetic SMPL-like body, procedural pose track, synthetic field
(see instantavatar_amd/synthetic.py): no dataset / checkpoint exists offline.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; frames are sharded
   across ranks, no data-path collective -> weak scaling)

Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--cpu-frames", type=int, default=2, help="frames timed on the CPU oracle (0 = skip)")
    ap.add_argument("--spinup-ms", type=float, default=600.0, help="untimed GPU clock spin-up before the warm-up steps")
    ap.add_argument("--no-profile", action="store_true", help="disable the in-library event timing")
    ap.add_argument("--no-graph", action="store_true", help="launch every frame eagerly instead of replaying the captured HIP "
                    "graph (eager: ~110 launches per frame from Python, a few hundred microseconds slower and jittery)")
    ap.add_argument("--train-steps", type=int, default=100, help="training iterations timed after the render loop (0 = skip)")
    return ap.parse_args()


def cpu_baseline(body, fp, model, poses, tr, res, n_frames):
    """The CPU restatement (oracle/) timed on this box's host cores on a bounded
    sample: `n_frames` full frames of the same workload (kind = "port")."""
    from oracle import oracle as orc
    from instantavatar_amd import synthetic as syn
    fd = model.deformer.deformer
    init = dict(tfs_inv_t=model.deformer.tfs_inv_t[0].cpu().numpy(),
                lbs_voxel=np.ascontiguousarray(fd.lbs_voxel_final[0].cpu().numpy()),
                offset_kernel=fd.offset_kernel.reshape(3).cpu().numpy().astype(np.float32),
                scale_kernel=fd.scale_kernel.reshape(3).cpu().numpy().astype(np.float32),
                bbox=model.deformer.bbox.cpu().numpy(), D=fd.resolution // 4, H=fd.resolution, W=fd.resolution)
    ro, rd = syn.make_camera_rays(res)
    rng = np.random.RandomState(0)
    orc.lib()
    t0 = time.time()
    for i in range(n_frames):
        world = orc.make_world(body, init, fp, np.zeros(10, np.float32), poses[i, 3:], poses[i, :3], tr[i], syn.INIT_BONES)
        jit = rng.rand(5, 64 ** 3, 3).astype(np.float32)
        orc.render_image_fast(world, ro, rd, jit)
    dt = time.time() - t0
    cores = os.cpu_count() or 1
    return {"value": n_frames / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d full %dx%d frames (occupancy build + render) through oracle/ (C + OpenMP, fp32)" % (n_frames, res, res),
            "seconds": dt}


def train_throughput(model, dev, poses, tr, rank, world_size, n_steps, res=512, n_rays=4096):
    """train.py analogue (configs 2/4): one frame + 4096 rays per step and rank
    (confs/sampler/patch.yaml: 4 x 32 x 32), targets rendered from the synthetic field,
    Adam(lr 1e-2), occupancy update every 20 steps, gradient all-reduce over RCCL."""
    from instantavatar_amd.pipeline import build_synthetic_model, make_batch
    from instantavatar_amd.training import NeRFLoss, configure_optimizer, training_step
    n_frames = 4
    targets = []
    with torch.no_grad():
        for f in range(n_frames):
            b = make_batch(dev, res, poses[f], tr[f])
            rgb, _, alpha, _ = model.render_image_fast(b, (res, res))
            targets.append((b, rgb.reshape(1, -1, 3), alpha.reshape(1, -1)))
    trainee, _, _ = build_synthetic_model(dev, resolution=128, n_levels=16)
    trainee.net_coarse.reset_parameters()   # identical seed on every rank -> identical replicas
    trainee.train()
    opt = configure_optimizer(trainee)
    loss_fn = NeRFLoss(dict(w_rgb=1.0, w_alpha=0.1, w_reg=0.1))
    g = torch.Generator(device=dev).manual_seed(1234 + rank)

    def step(i):
        b, rgb, alpha = targets[(i + rank) % n_frames]
        sel = torch.randint(0, res * res, (n_rays,), device=dev, generator=g)
        batch = dict(b)
        for k in ("rays_o", "rays_d"):
            batch[k] = b[k][:, sel]
        for k in ("near", "far"):
            batch[k] = b[k][:, sel]
        batch["rgb"], batch["alpha"] = rgb[:, sel], alpha[:, sel]
        batch["bg_color"] = torch.ones_like(batch["rgb"])
        return training_step(trainee, batch, opt, loss_fn, world_size=world_size)

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    if world_size > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    first = last = None
    for i in range(n_steps):
        out = step(3 + i)
        if i == 0:
            first = out["mse_loss"].detach()
        last = out["mse_loss"].detach()
    torch.cuda.synchronize()
    if world_size > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world_size > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return {"it_per_sec": n_steps / dt, "rays_per_sec": n_steps * n_rays * world_size / dt, "steps": n_steps,
            "rays_per_step_per_gpu": n_rays, "mse_first": float(first), "mse_last": float(last),
            "note": "global batch = n_gpus x 4096 rays (weak scaling); occupancy update every 20 steps included"}


def hashgrid_roofline(model, dev, n=1 << 20, reps=20):
    """The hash-grid lookup in isolation (north_star: fraction of the HBM roofline on the hash-grid
    lookup): the XCD-sharded encoding kernel on 2^20 uniformly random points of the field's bounding
    box, timed with events on the launch stream.  Algorithmic bytes = 512 B per sample."""
    net = model.net_coarse
    bb = model.deformer.bbox
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.rand((n, 3), device=dev, generator=g) * (bb[1] - bb[0]) + bb[0]
    with torch.no_grad():
        for _ in range(3):
            net.encode_planes(x)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            net.encode_planes(x)
        b.record()
        torch.cuda.synchronize()
    us = a.elapsed_time(b) / reps * 1e3
    gbs = n * 512 / (us * 1e-6) / 1e9
    out = {"kernel": "k_encode_xcd", "samples": n, "avg_launch_us": us, "Gsamples_per_s": n / us * 1e-3, "bound": "hbm",
           "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
           "note": ("the 26 MB fp16 table is served from the per-XCD L2s (one hashed level per XCD): the binding limit is the "
                    "L2 request rate, 8 XCD x 16 channels x 2.1 GHz = 269 G requests/s; see l2_* (profiles/r01_pmc_encode.json)")}
    pj = os.path.join(ROOT, "profiles", "r01_pmc_encode.json")
    if os.path.exists(pj):
        try:
            c = json.load(open(pj))["k_encode_xcd<16>"]
            req = c["l2_read_requests_per_sample"]
            out.update(l2_hit_rate=c["l2_hit_rate"], l2_read_requests_per_sample=req,
                       l2_request_rate_frac=req * n / (us * 1e-6) / 269e9,
                       traffic=c["fabric_fetch_bytes_per_launch_x2_corrected"])
        except Exception:
            pass
    return out


def main():
    args = parse()
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world_size > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from instantavatar_amd import _lib, synthetic as syn
    from instantavatar_amd.pipeline import build_synthetic_model, make_batch

    torch.manual_seed(42 + rank)
    model, body, fp = build_synthetic_model(dev, resolution=128, n_levels=16)
    res = args.res
    n_total = args.steps + args.warmup
    poses, tr = syn.procedural_pose_track(max(200, n_total * world_size))
    # frames are sharded round-robin over ranks (config 3: embarrassingly parallel)
    my = [rank + i * world_size for i in range(n_total)]
    batches = [make_batch(dev, res, poses[f % len(poses)], tr[f % len(poses)]) for f in my[:8]]
    # share the (identical) camera rays between batches: one resident copy
    for b in batches[1:]:
        b["rays_o"], b["rays_d"] = batches[0]["rays_o"], batches[0]["rays_d"]
    pose_t = torch.as_tensor(poses, device=dev)
    tr_t = torch.as_tensor(tr, device=dev)

    def frame(i):
        f = my[i] % len(poses)
        b = batches[i % len(batches)]
        b["global_orient"], b["body_pose"], b["transl"] = pose_t[f:f + 1, :3], pose_t[f:f + 1, 3:], tr_t[f:f + 1]
        d = float(np.sqrt((tr[f] ** 2).sum()))
        b["near"].fill_(d - 1)
        b["far"].fill_(d + 1)
        return model.render_image_fast(b, (res, res))

    L = _lib.lib()
    eager_frame = frame
    graphed = None
    mode = "eager"
    if not args.no_graph:
        try:
            from instantavatar_amd.pipeline import GraphedRenderer
            # iteration count of the wave-front loop measured on 12 poses spread over this rank's frames
            probes = []
            for j in range(12):
                f = my[(j * max(n_total // 12, 1)) % n_total] % len(poses)
                pb = dict(batches[0])
                d = float(np.sqrt((tr[f] ** 2).sum()))
                pb["global_orient"], pb["body_pose"], pb["transl"] = pose_t[f:f + 1, :3], pose_t[f:f + 1, 3:], tr_t[f:f + 1]
                pb["near"], pb["far"] = torch.full_like(batches[0]["near"], d - 1), torch.full_like(batches[0]["far"], d + 1)
                probes.append(pb)
            graphed = GraphedRenderer(model, batches[0], (res, res), margin=2, probe_batches=probes)

            def frame(i):  # noqa: F811  (same work, replayed from the captured HIP graph)
                f = my[i] % len(poses)
                d = float(np.sqrt((tr[f] ** 2).sum()))
                b = batches[0]
                b["global_orient"], b["body_pose"], b["transl"] = pose_t[f:f + 1, :3], pose_t[f:f + 1, 3:], tr_t[f:f + 1]
                b["near"].fill_(d - 1)
                b["far"].fill_(d + 1)
                return graphed(b)
            mode = "hip_graph"
        except Exception as e:  # capture not possible on this stack: stay eager (still the HIP path)
            print("graph capture failed, running eagerly:", repr(e)[:200], file=sys.stderr)
            frame = eager_frame
    # clock spin-up: a GPU that idled through model construction needs a few hundred ms of load to
    # reach its sustained clocks (measured: 229 vs 296 frames/s with 45 vs 400 frames run);
    # these frames are neither warm-up nor timed steps and are reported as `spinup_ms`
    t_spin = time.perf_counter()
    while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms:
        frame(0)
        torch.cuda.synchronize()
    for i in range(args.warmup):
        out = frame(i)
    torch.cuda.synchronize()
    if world_size > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cnt_sum = torch.zeros((), device=dev)
    cov_sum = torch.zeros((), device=dev)
    for i in range(args.warmup, n_total):
        rgb, depth, alpha, counter = frame(i)
        cnt_sum += counter.mean()
        cov_sum += (alpha > 0.5).float().mean()
    torch.cuda.synchronize()
    if world_size > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # frames whose wave-front loop needed more iterations than the graph holds (deferred check) are
    # rendered again eagerly; that time belongs to the job
    incomplete = graphed.finish() if graphed is not None else 0
    if incomplete:
        redo = [c for c in graphed.incomplete_calls]
        t_re = time.perf_counter()
        for _ in redo:
            eager_frame(args.warmup)
        torch.cuda.synchronize()
        dt += time.perf_counter() - t_re
    # per-kernel timing for the roofline block: HIP events around every launch of the two dominant
    # kernels on the stream they run on.  Recording ~50 events per frame costs ~10 % of the frame,
    # so `value` comes from the un-instrumented pass above and the SAME K frames are then launched
    # once more with the events enabled (`ms_per_step_instrumented`).
    prof = not args.no_profile
    dt_prof = None
    if prof:
        _lib.check(L.ia_profile_enable(1))
        _lib.check(L.ia_profile_reset())
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        for i in range(args.warmup, n_total):
            eager_frame(i)
        torch.cuda.synchronize()
        dt_prof = time.perf_counter() - tp0
    if world_size > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    roof = None
    kernels = {}
    if prof:
        for kid, name in ((0, "k_search"), (1, "k_field")):
            ms, n, units = C.c_double(), C.c_int64(), (C.c_uint64 * 2)()
            _lib.check(L.ia_profile_get(kid, C.byref(ms), C.byref(n), units))
            kernels[name] = dict(ms=ms.value, launches=n.value, units=[int(units[0]), int(units[1])])
        _lib.check(L.ia_profile_enable(0))
        ks, kf = kernels["k_search"], kernels["k_field"]
        # algorithmic bytes: field = 512 B gathered (16 lv x 8 corners x 2 x fp16)
        # + 12 B in + 16 B out per sample; search = 384 B per trilinear fetch (8 corners x 12 ch
        # x 4 B) + 12 B in per point + 12 B out per surviving root (bounded by solves).
        kf["bytes"] = kf["units"][0] * (512 + 12 + 16)
        ks["bytes"] = ks["units"][1] * 384 + ks["units"][0] // 13 * 12
        dom = "k_field" if kf["ms"] >= ks["ms"] else "k_search"
        k = kernels[dom]
        per_launch_ms = k["ms"] / max(k["launches"], 1)
        achieved = k["bytes"] / max(k["launches"], 1) / (per_launch_ms * 1e-3) / 1e9 if k["ms"] > 0 else 0.0
        traffic, traffic_src = None, None
        tp = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tp):  # HBM bytes per launch from the committed PMC passes (tools/pmc_traffic.py)
            try:
                tj = json.load(open(tp))
                traffic, traffic_src = tj[dom]["hbm_bytes_per_launch"], "profiles/r01_pmc_traffic.json"
            except Exception:
                pass
        roof = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "note": ("algorithmic bytes: k_search = 384 B per trilinear fetch of the 25 MB transform grid (L2 / Infinity-Cache "
                         "resident, so achieved can exceed the HBM peak; `traffic` is what reached the fabric); "
                         "k_field (encode + MLP kernels) = 540 B per sample (512 B hash-table gathers).  The fetch pattern of "
                         "k_search measured in isolation (tools/ubench/records.hip, 24 x 16-byte loads per lane and fetch): "
                         "35.7 G fetches/s with L1-resident cells, 23.8 G/s L2-resident, 11.0 G/s from the fabric"),
                "fetches_per_s": ks["units"][1] / (ks["ms"] * 1e-3) if ks["ms"] > 0 else 0.0,
                "fetch_pattern_ceiling_per_s": {"l1_resident": 35.7e9, "l2_resident": 23.8e9, "fabric": 11.0e9},
                # the fraction that says something about the kernel: fetch rate vs the measured ceiling of its own
                # access pattern with cache-resident cells (`frac` above exceeds 1 because nothing comes from HBM)
                "frac_of_fetch_ceiling": (ks["units"][1] / (ks["ms"] * 1e-3) / 35.7e9) if ks["ms"] > 0 else 0.0,
                "avg_launch_us": per_launch_ms * 1e3, "launches": k["launches"],
                "algorithmic_bytes_per_launch": k["bytes"] / max(k["launches"], 1),
                "other": {n: {"avg_launch_us": v["ms"] * 1e3 / max(v["launches"], 1), "launches": v["launches"],
                              "GBps": (v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0.0),
                              "units": v["units"]} for n, v in kernels.items()}}

    frames = args.steps * world_size
    fps = frames / dt
    result = {
        "metric": "novel_pose_render_frames_per_sec_512x512", "value": fps, "unit": "frames/s",
        "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "render_image_fast %dx%d, SNARF_NGP defaults (13 init bones, 128^2x32 skinning voxels, "
                               "16-level hash grid T=2^19, 64^3 occupancy rebuilt per frame with 5 probes, "
                               "MAX_SAMPLES 256, MAX_BATCH 291600), synthetic SMPL-like body + procedural poses" % (res, res),
                   "frames_sharded_over": world_size},
        "rays_per_sec": fps * res * res,
        "samples_per_ray": float(cnt_sum.item()) / args.steps,
        "alpha_coverage": float(cov_sum.item()) / args.steps,
        "render_loop_iters": model.renderer.last_iters, "launch_mode": mode, "spinup_ms": args.spinup_ms,
        "frames_rerendered_eagerly": int(incomplete),
        "ms_per_step_instrumented": (dt_prof / args.steps * 1e3) if dt_prof else None,
    }
    if roof is not None:
        result["roofline"] = roof
    if rank == 0 and prof:
        result["hashgrid_lookup"] = hashgrid_roofline(model, dev)
    if args.train_steps > 0:
        result["train"] = train_throughput(model, dev, poses, tr, rank, world_size, args.train_steps, res=res)
    if rank == 0 and world_size == 1 and args.cpu_frames > 0:
        result["cpu_baseline"] = cpu_baseline(body, fp, model, poses, tr, res, args.cpu_frames)
    if rank == 0:
        print(json.dumps(result))
    if world_size > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
