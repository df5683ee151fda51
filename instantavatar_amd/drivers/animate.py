"""`animate.py` without Lightning / Hydra (reference animate.py:13-118): render a novel-pose sequence
with a trained (or synthetic) avatar and write PNG frames + a GIF.

    python -m instantavatar_amd.drivers.animate --poses data/animation/aist_demo.npz \\
        --ckpt checkpoints/last.ckpt --smpl-dir ./data/SMPLX/smpl --gender male --out animation/aist_demo
    python -m instantavatar_amd.drivers.animate --synthetic --max-frames 8 --downscale 8 --out /tmp/anim

Camera, ray construction and the pose-track conventions are the reference's AnimateDataset
(animate.py:13-80): 1080x1080 pinhole with f = 2000, c2w = I, `trans - trans[0] + (0, 0.15, 5)`,
near/far = |transl| -+ 1.  Frames are replayed from one captured HIP graph (pipeline.GraphedRenderer)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

from .. import synthetic
from ..pipeline import AvatarModel, build_synthetic_model
from . import checkpoint as ckpt_io
from . import config as cfg


def make_rays(K, c2w, H, W):
    """animate.py:13-25"""
    x, y = np.meshgrid(np.arange(W), np.arange(H), indexing="xy")
    xy = np.stack([x, y, np.ones_like(x)], axis=-1).reshape(-1, 3).astype(np.float32)
    d_c = xy @ np.linalg.inv(K).T
    d_w = d_c @ c2w[:3, :3].T
    d_w = d_w / np.linalg.norm(d_w, axis=1, keepdims=True)
    o_w = np.tile(c2w[:3, 3], (len(d_w), 1))
    return o_w.astype(np.float32), d_w.astype(np.float32)


class AnimateSequence:
    """AnimateDataset (animate.py:27-80) as device-resident batches."""

    def __init__(self, poses72, trans, betas, device, downscale=2, size=None):
        """size: a square image of that edge instead of 1080 // downscale (the same camera scaled by size / 1080: the 512 x 512
        frame BASELINE.json's metric is quoted on)."""
        H = W = 1080
        K = np.eye(3)
        K[0, 0] = K[1, 1] = 2000
        K[0, 2] = H // 2
        K[1, 2] = W // 2
        if size:
            K[:2] *= size / H
            H = W = int(size)
        elif downscale > 1:
            H, W = H // downscale, W // downscale
            K[:2] /= downscale
        self.H, self.W = H, W
        o, d = make_rays(K, np.eye(4), H, W)
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device)
        self.rays_o, self.rays_d = t(o)[None], t(d)[None]
        self.thetas = t(poses72[..., :72])
        self.transl = t(trans - trans[0:1] + np.array([0, 0.15, 5], np.float32))
        self.betas = t(betas).reshape(1, 10)

    def __len__(self):
        return self.transl.shape[0]

    def batch(self, idx, rays=True):
        """rays=False: the SMPL parameters only (views, no launch) -- all a captured frame graph reads per frame; its rays
        are static and near / far are recomputed from the ray origins inside the renderer (snarf_deformer.py:101-103)."""
        pose = {"betas": self.betas, "global_orient": self.thetas[idx:idx + 1, :3], "body_pose": self.thetas[idx:idx + 1, 3:],
                "transl": self.transl[idx:idx + 1]}
        if not rays:
            return pose
        dist = torch.sqrt((self.transl[idx] ** 2).sum())
        ones = torch.ones(1, self.rays_d.shape[1], device=self.rays_d.device)
        return {"rays_o": self.rays_o, "rays_d": self.rays_d, **pose, "near": ones * (dist - 1), "far": ones * (dist + 1)}


def build_model(args, device, quiet=False):
    if args.synthetic:
        model, _, _ = build_synthetic_model(device, seed=int(getattr(args, "seed", None) or 42))
        if args.ckpt:  # synthetic body, trained weights (e.g. written by drivers.train --synthetic)
            missing, unexpected = ckpt_io.load_checkpoint(model, args.ckpt, map_location=device, strict_self_check=True)
            if not quiet:
                print("checkpoint %s loaded (step %d)" % (args.ckpt, model.global_step))
        return model, np.zeros(10, np.float32)
    deformer, net, renderer = cfg.build_plugins(args.confs, args.deformer, args.network, args.renderer, gender=args.gender,
                                                deformer_kwargs=dict(model_path=args.smpl_dir))
    model = AvatarModel(deformer, net, renderer).to(device)
    renderer.initialize(1)
    betas = np.zeros(10, np.float32)
    if args.betas:
        betas = np.load(args.betas)["betas"].reshape(-1)[:10].astype(np.float32)
    missing, unexpected = ckpt_io.load_checkpoint(model, args.ckpt, map_location=device, strict_self_check=True)
    if not quiet:
        print("checkpoint %s: %d tensors not on the path ignored, %d own tensors kept at init" % (args.ckpt, len(unexpected), len(missing)))
    if not getattr(deformer, "initialized", False):
        deformer.initialize(torch.as_tensor(betas, device=device).reshape(1, 10), device)
        deformer.initialized = True
    # The reference's animate.py never calls net.initialize: `center` / `scale` come from the checkpoint
    # (buffers written by training, DNeRF.py:134).  Only a checkpoint without them falls back to the bbox.
    if "net_coarse.center" in missing or "net_coarse.scale" in missing:
        net.initialize(deformer.bbox)
    else:
        net.bbox = deformer.bbox    # attribute only (NeRFNGPNet.initialize is a no-op once `bbox` exists)
    return model, betas


def _bgra_to_rgba(f):
    return np.ascontiguousarray(np.asarray(f)[..., [2, 1, 0, 3]])


def write_frames(frames, out_dir, gif=None, indices=None, workers=8):
    """8-bit [H, W, 4] frames in the MODEL's channel order -> `<i>.png` (+ optionally one GIF) with the colours the reference's
    files have.  The reference writes with cv2.imwrite (animate.py:113), which takes channel 0 as BLUE: the model's channels are
    in the (B, G, R) order of the cv2.imread training images (peoplesnapshot.py:100), so its files hold the intended colours;
    for the GIF it converts BGRA -> RGBA first (:115).  PIL takes channel 0 as RED: the same reorder serves both.
    indices: the file number of every frame (a rank of a multi-GPU job writes only the frames it rendered); PNG encoding
    runs on `workers` threads (zlib releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    os.makedirs(out_dir, exist_ok=True)
    indices = list(range(len(frames))) if indices is None else list(indices)

    def one(job):
        i, f = job
        Image.fromarray(_bgra_to_rgba(f), "RGBA").save(os.path.join(out_dir, "%d.png" % i))
    with ThreadPoolExecutor(max_workers=max(1, workers)) as ex:
        list(ex.map(one, zip(indices, frames)))
    if gif and len(frames):
        write_gif(frames, os.path.join(out_dir, gif))


def write_gif(frames, path):
    from PIL import Image
    ims = [Image.fromarray(_bgra_to_rgba(f), "RGBA") for f in frames]
    ims[0].save(path, save_all=True, append_images=ims[1:], duration=33, loop=0)


def pack_rgba8(out, dst):
    """(rgb [1,H,W,3], depth, alpha [1,H,W], counter) of one frame -> `dst` uint8 [H,W,4] on the device: animate.py:107-113's
    `cat([rgb, alpha[..., None]]) * 255 -> uint8` as ONE launch on the current stream (`ia_pack_rgba8`)."""
    from .. import _lib
    rgb, _, alpha, _ = out
    _lib.require_cuda(rgb, alpha, dst)
    _lib.check(_lib.lib().ia_pack_rgba8(_lib.ptr(rgb), _lib.ptr(alpha), alpha.numel(), _lib.ptr(dst), _lib.stream()), "ia_pack_rgba8")


def fixed_jitter(seed, device, iters=5, G=64):
    """One occupancy-probe jitter for the whole sequence (density_grid.py:98 draws a fresh torch.rand per frame): with it a
    frame is a function of its pose alone, so a sequence rendered by 1, 2 or 8 ranks gives the same files."""
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    return torch.rand((iters, G * G * G, 3), generator=g).to(device)


def _make_renderer(model, first, size, in_flight, probes, jitter):
    from ..pipeline import PipelinedRenderer
    return PipelinedRenderer(model, first, size, n_in_flight=in_flight, margin=1, probe_batches=probes, jitter=jitter)


def render_sequence(model, seq, out_dir, gif="animation.gif", launch=None, in_flight=3, jitter=None, make_renderer=None, log=print,
                    max_buffer_bytes=1 << 30):
    """animate.py:104-118 / novel_view.py:117-127 for one rank of a job of `launch.world_size` ranks.

    The frames of `seq` are dealt round-robin to the ranks (parallel.shard_frames; frames are independent: no data-path
    collective).  A rank renders its frames through `pipeline.PipelinedRenderer`: `in_flight` captured HIP graphs replayed
    round-robin on their own streams, so a second frame fills the gaps the latency-bound tail of the first leaves.  Behind
    every frame, on the frame's stream, the image is packed to 8-bit RGBA on the device (pack_rgba8) and copied to pinned
    host memory asynchronously: the loop never waits for the GPU.  Frames whose wave-front loop needed more iterations than
    the graph holds are rendered again eagerly.  Every rank writes its own `<i>.png`; the GIF is written by rank 0 from the
    frames gathered there (RCCL gather of the packed images).
    Returns a dict: frames (of the sequence), local (rendered here), render_s (this rank's render loop, I/O and graph capture
    excluded), frames_per_sec (whole job: frames / max over ranks of render_s), incomplete (re-rendered frames)."""
    from .launch import Launch
    from ..parallel import shard_frames
    launch = launch or Launch(device=seq.rays_o.device)
    size = (seq.H, seq.W)
    dev = seq.rays_o.device
    on_gpu = dev.type == "cuda"
    mine = shard_frames(len(seq), launch.rank, launch.world_size)
    n = len(mine)
    # the packed frames stay on the device (the GIF gather travels over RCCL from there) and in pinned host memory (PNG encoding).
    # With a GIF every frame is kept until the end, as the reference does (animate.py:106-116); without one the sequence goes
    # through buffers of at most `max_buffer_bytes` (render a chunk, write its files, reuse the buffers)
    per_frame = seq.H * seq.W * 4
    cap = n if gif else max(1, min(n, int(max_buffer_bytes) // per_frame))
    dev_frames = torch.empty((cap, seq.H, seq.W, 4), dtype=torch.uint8, device=dev)
    host = torch.empty((cap, seq.H, seq.W, 4), dtype=torch.uint8, pin_memory=on_gpu)
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    render_s, redo = 0.0, []
    if n:
        probes = [seq.batch(mine[(j * max(n // 8, 1)) % n]) for j in range(min(8, n))]   # wave-front loop length over the shard
        renderer = (make_renderer or _make_renderer)(model, seq.batch(mine[0]), size, max(1, min(in_flight, n)), probes, jitter)

        # device -> host copies run on their own stream behind an event: a frame's stream goes straight on to its next frame
        # instead of standing still for the PCIe transfer (every frame of a chunk has its own slot: nothing is reused in flight)
        copier = torch.cuda.Stream(device=dev) if on_gpu else None

        def keep(slot):
            def consume(out, k):
                pack_rgba8(out, dev_frames[slot])
                if copier is None:
                    host[slot].copy_(dev_frames[slot])
                    return
                ev = torch.cuda.Event()
                ev.record()
                copier.wait_event(ev)
                with torch.cuda.stream(copier):
                    host[slot].copy_(dev_frames[slot], non_blocking=True)
            return consume
        with torch.inference_mode():
            for c0 in range(0, n, cap):
                c1 = min(n, c0 + cap)
                sync()
                t0 = time.perf_counter()
                for j in range(c0, c1):
                    renderer(seq.batch(mine[j], rays=False), keep(j - c0))
                renderer.synchronize()
                if copier is not None:
                    copier.synchronize()
                render_s += time.perf_counter() - t0
                renderer.finish()
                bad = [c for c in renderer.incomplete_calls if c0 <= c < c1]     # (call number = position in `mine`)
                for c in bad:
                    keep(c - c0)(model.render_image_fast(seq.batch(mine[c]), size, jitter=jitter), 0)
                sync()
                redo += bad
                if not gif:
                    write_frames(host[:c1 - c0].numpy(), out_dir, indices=mine[c0:c1])
    slowest = launch.max_over_ranks(render_s)
    if gif:
        write_frames(host.numpy(), out_dir, indices=mine)
        parts = launch.gather_to_main(dev_frames if launch.backend == "nccl" else host)
        if launch.is_main:
            w = launch.world_size
            write_gif([parts[i % w][i // w].numpy() for i in range(len(seq))], os.path.join(out_dir, gif))
    launch.barrier()
    res = dict(frames=len(seq), local=n, render_s=render_s, frames_per_sec=(len(seq) / slowest if slowest > 0 else 0.0),
               incomplete=len(redo), in_flight=in_flight, world_size=launch.world_size)
    if log is not None and launch.is_main:
        log("rendered %d frames (%dx%d) in %.3f s = %.1f frames/s (%d rank(s), %d in flight per GPU, PNG / GIF I/O and graph capture "
            "excluded; %d re-rendered eagerly)" % (res["frames"], seq.W, seq.H, slowest, res["frames_per_sec"], launch.world_size, in_flight, len(redo)))
    return res


def add_launch_args(ap):
    ap.add_argument("--in-flight", type=int, default=3, help="frames in flight per GPU (captured HIP graphs replayed round-robin on their own streams)")
    ap.add_argument("--jitter-seed", type=int, default=None,
                    help="use ONE occupancy-probe jitter for all frames (drawn from this seed) instead of a fresh draw per frame "
                         "(density_grid.py:98): the files no longer depend on how many ranks rendered the sequence")


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--poses", help="npz with `poses` [n,>=72] and `trans` [n,3] (data/animation/aist_demo.npz)")
    ap.add_argument("--ckpt", help="Lightning checkpoint of DNeRFModel")
    ap.add_argument("--betas", help="npz with `betas` (the subject's anim_nerf_train.npz)")
    ap.add_argument("--smpl-dir", default="./data/SMPLX/smpl")
    ap.add_argument("--gender", default="neutral")
    ap.add_argument("--confs", default=os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "confs"))
    ap.add_argument("--deformer", default="fast_snarf")
    ap.add_argument("--network", default="ngp")
    ap.add_argument("--renderer", default="raymarcher_acc")
    ap.add_argument("--synthetic", action="store_true", help="synthetic SMPL-like body + field (no SMPL pickle / checkpoint needed)")
    ap.add_argument("--downscale", type=int, default=2)
    ap.add_argument("--size", type=int, default=0, help="square image edge in pixels (overrides --downscale), e.g. 512")
    ap.add_argument("--max-frames", type=int, default=0)
    ap.add_argument("--out", default="animation/out")
    ap.add_argument("--no-gif", action="store_true")
    ap.add_argument("--subjects", help="BASELINE config 5 (a batch of subjects, one per GPU): a JSON list of per-subject overrides "
                    "[{\"ckpt\": ..., \"betas\": ..., \"gender\": ..., \"out\": ..., (\"poses\": ..., \"seed\": ...)}, ...]; rank r of the job renders "
                    "the WHOLE sequence of subject r (r modulo the list) with that subject's weights into its `out` -- independent "
                    "replicas, no frame sharding, no data-path collective")
    ap.add_argument("--seed", type=int, default=None, help="--synthetic: seed of the synthetic body + field (another seed = another subject)")
    add_launch_args(ap)
    args = ap.parse_args(argv)
    from .launch import Launch
    launch = Launch.from_env(who="animate")
    device = launch.device
    replica = None
    if args.subjects:
        import json
        with open(args.subjects) as f:
            subjects = json.load(f)
        if not isinstance(subjects, list) or not subjects:
            ap.error("--subjects: a non-empty JSON list of objects")
        mine = subjects[launch.rank % len(subjects)]
        unknown = set(mine) - {"ckpt", "betas", "gender", "out", "poses", "seed", "smpl_dir"}
        if unknown:
            ap.error("--subjects: unknown keys %s" % sorted(unknown))
        for k, v in mine.items():
            setattr(args, k, v)
        if "out" not in mine:
            args.out = os.path.join(args.out, "subject_%d" % (launch.rank % len(subjects)))
        replica = Launch(device=device)     # this rank's own one-rank "job": render_sequence deals it every frame
    if not args.synthetic and not (args.ckpt and args.poses):
        ap.error("--ckpt and --poses are required unless --synthetic is given")
    try:
        model, betas = build_model(args, device, quiet=not launch.is_main)
        model.eval()
        track = args.poses
        if not track and args.max_frames > 64:   # a longer synthetic run: the pose track the reference ships (its first 200 frames travel with the tests)
            track = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "aist_demo_200.npz")
        if track:
            z = np.load(track)
            poses, trans = z["poses"].astype(np.float32), z["trans"].astype(np.float32)
        else:
            poses, trans = synthetic.procedural_pose_track(64)
        if args.max_frames:
            poses, trans = poses[:args.max_frames], trans[:args.max_frames]
        seq = AnimateSequence(poses, trans, betas, device, args.downscale, size=args.size or None)
        jitter = None if args.jitter_seed is None else fixed_jitter(args.jitter_seed, device)
        res = render_sequence(model, seq, args.out, gif=None if args.no_gif else "animation.gif", launch=replica or launch,
                              in_flight=args.in_flight, jitter=jitter, log=None if replica else print)
        if replica is not None:
            # whole-job figure of the replicas: all frames of all subjects over the slowest rank's render loop
            slowest = launch.max_over_ranks(res["render_s"])
            total = launch.sum_over_ranks(res["frames"])
            print("rank %d: subject %d, %d frames (%dx%d) to %s" % (launch.rank, launch.rank % len(subjects), res["frames"], seq.W, seq.H, args.out))
            launch.barrier()
            if launch.is_main:
                print("rendered %d frames of %d subject replica(s) in %.3f s = %.1f frames/s (one subject per rank; PNG / GIF I/O and graph capture excluded)"
                      % (int(total), launch.world_size, slowest, total / slowest if slowest > 0 else 0.0))
        elif launch.is_main:
            print("wrote %d frames (%dx%d) to %s" % (res["frames"], seq.W, seq.H, args.out))
    finally:
        launch.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
