"""Generates tests/golden/pose_tracks.npz: the SMPL parameter tracks BASELINE configs 2 and 4 name -- the reference's
data/PeopleSnapshot/male-3-casual/poses/anim_nerf_train.npz (114 frames, what PeopleSnapshotDataset serves as
`betas` / `global_orient` / `body_pose` / `transl`, peoplesnapshot.py:127-131) and data/custom/seattle/poses/train.npz
(41 frames, the Neuman sequence of SNARF_NGP_refine.yaml).  A data fixture (raw arrays, ~50 KB), not code; the tests
render frames of these tracks with the bench model against the oracle (tests/test_gpu_fullconfig.py).
Run from the repo root:  python tests/golden/make_pose_tracks.py"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = {"male3": "/root/reference/data/PeopleSnapshot/male-3-casual/poses/anim_nerf_train.npz",
       "seattle": "/root/reference/data/custom/seattle/poses/train.npz"}
out = {}
for name, path in SRC.items():
    z = np.load(path)
    for k in ("betas", "global_orient", "body_pose", "transl"):
        out["%s_%s" % (name, k)] = z[k].astype(np.float32)
p = os.path.join(HERE, "pose_tracks.npz")
np.savez_compressed(p, **out)
print("wrote", p, os.path.getsize(p), "bytes", {k: v.shape for k, v in out.items()})
