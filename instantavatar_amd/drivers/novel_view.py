"""`novel_view.py` without Lightning / Hydra (reference novel_view.py:27-131): the avatar in a fixed canonical-like pose, turned
once around its vertical axis in `--frames` steps (60 in the reference), rendered at 1080 / downscale and written as
PNG frames + a GIF.

    python -m instantavatar_amd.drivers.novel_view --ckpt checkpoints/last.ckpt --smpl-dir ./data/SMPLX/smpl --out animation/rotation
    python -m instantavatar_amd.drivers.novel_view --synthetic --frames 8 --downscale 8 --out /tmp/rot

Camera and rays are animate.py's (1080 x 1080 pinhole, f = 2000, c2w = I); the pose is novel_view.py:45-49
(global_orient (pi, 0, 0), body_pose zero except entries 2 / 5 = +-0.5, transl (0, 0.5, 5)); frame i turns the body by
2 pi i / n about y: R_y(angle) @ R(global_orient), converted back to a rotation vector (novel_view.py:80-85, cv2.Rodrigues
both ways).  The batch's near / far (0 / 10 in the reference, :88-89) are replaced by |o| -+ 1 inside the renderer
(transform_rays_w2s), exactly as in the reference."""
import argparse
import os
import sys

import numpy as np
import torch

from .animate import add_launch_args, build_model, fixed_jitter, make_rays, render_sequence


def rotvec_to_matrix(v):
    """Rodrigues' formula (cv2.Rodrigues, vector -> matrix), float64."""
    v = np.asarray(v, np.float64).reshape(3)
    th = np.linalg.norm(v)
    if th < 1e-12:
        return np.eye(3)
    k = v / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def matrix_to_rotvec(R):
    """cv2.Rodrigues, matrix -> vector: angle in [0, pi].  Away from pi the axis is the skew part / sin; towards pi the skew
    part vanishes, and the axis comes from the symmetric part, (R + R^T) / 2 = cos I + (1 - cos) k k^T, with its sign fixed
    by the skew part where that is non-zero (at exactly pi the two signs describe the same rotation)."""
    R = np.asarray(R, np.float64)
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    s = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2      # sin(th) k
    th = np.arctan2(np.linalg.norm(s), c)
    if th < 1e-12:
        return np.zeros(3)
    if th < 2.5:
        return s / np.linalg.norm(s) * th
    M = ((R + R.T) / 2 - c * np.eye(3)) / (1 - c)                                    # k k^T
    i = int(np.argmax(np.diag(M)))
    k = M[i] / np.sqrt(M[i, i])
    if np.abs(s).max() > 1e-300 and np.dot(k, s) < 0:
        k = -k
    return k / np.linalg.norm(k) * th


class RotationSequence:
    """novel_view.py:27-91 (`AnimateDataset`) as device-resident batches."""

    def __init__(self, num_frames, betas, device, downscale=2):
        H = W = 1080
        K = np.eye(3)
        K[0, 0] = K[1, 1] = 2000
        K[0, 2] = H // 2
        K[1, 2] = W // 2
        if downscale > 1:
            H, W = H // downscale, W // downscale
            K[:2] /= downscale
        self.H, self.W = H, W
        o, d = make_rays(K, np.eye(4), H, W)
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device)
        self.rays_o, self.rays_d = t(o)[None], t(d)[None]
        body_pose = np.zeros((1, 69), np.float32)
        body_pose[:, 2], body_pose[:, 5] = 0.5, -0.5
        self.body_pose = t(body_pose)
        self.transl = t(np.array([[0, 0.5, 5]], np.float32))
        self.betas = t(np.asarray(betas, np.float32)).reshape(1, 10)
        R0 = rotvec_to_matrix(np.array([np.pi, 0, 0]))
        orient = [matrix_to_rotvec(rotvec_to_matrix(np.array([0, 2 * np.pi * i / num_frames, 0])) @ R0) for i in range(num_frames)]
        self.global_orient = t(np.stack(orient).astype(np.float32))
        self.num_frames = num_frames

    def __len__(self):
        return self.num_frames

    def batch(self, idx, rays=True):
        """rays=False: the SMPL parameters only (see animate.AnimateSequence.batch)"""
        pose = {"betas": self.betas, "global_orient": self.global_orient[idx:idx + 1], "body_pose": self.body_pose, "transl": self.transl}
        if not rays:
            return pose
        ones = torch.ones(1, self.rays_d.shape[1], device=self.rays_d.device)
        return {"rays_o": self.rays_o, "rays_d": self.rays_d, **pose, "near": ones * 0, "far": ones * 10}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--ckpt", help="Lightning checkpoint of DNeRFModel")
    ap.add_argument("--betas", help="npz with `betas` (the subject's anim_nerf_train.npz)")
    ap.add_argument("--smpl-dir", default="./data/SMPLX/smpl")
    ap.add_argument("--gender", default="neutral")
    ap.add_argument("--confs", default=os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "confs"))
    ap.add_argument("--deformer", default="fast_snarf")
    ap.add_argument("--network", default="ngp")
    ap.add_argument("--renderer", default="raymarcher_acc")
    ap.add_argument("--synthetic", action="store_true", help="synthetic SMPL-like body + field (no SMPL pickle / checkpoint needed)")
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--downscale", type=int, default=2)
    ap.add_argument("--out", default="animation/rotation")
    ap.add_argument("--no-gif", action="store_true")
    add_launch_args(ap)
    args = ap.parse_args(argv)
    if not args.synthetic and not args.ckpt:
        ap.error("--ckpt is required unless --synthetic is given")
    from .launch import Launch
    launch = Launch.from_env(who="novel_view")
    device = launch.device
    try:
        model, betas = build_model(args, device, quiet=not launch.is_main)
        model.eval()
        seq = RotationSequence(args.frames, betas, device, args.downscale)
        jitter = None if args.jitter_seed is None else fixed_jitter(args.jitter_seed, device)
        res = render_sequence(model, seq, args.out, gif=None if args.no_gif else "rotation.gif", launch=launch,
                              in_flight=args.in_flight, jitter=jitter)
        if launch.is_main:
            print("wrote %d frames (%dx%d) to %s" % (res["frames"], seq.W, seq.H, args.out))
    finally:
        launch.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
