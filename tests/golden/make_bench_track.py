"""Generates tests/golden/aist_demo_200.npz: the pose track BASELINE config 3 / SURVEY 8(d) names for the headline bench --
the first 200 frames of the reference's data/animation/aist_demo.npz (`poses[:, :72]`, `trans`), raw; the bench applies
animate.py:48-50 (`trans - trans[0] + (0, 0.15, 5)`).  A data fixture (58 KB), not code.
Run from the repo root:  python tests/golden/make_bench_track.py"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
z = np.load("/root/reference/data/animation/aist_demo.npz")
out = os.path.join(HERE, "aist_demo_200.npz")
np.savez_compressed(out, poses=z["poses"][:200, :72].astype(np.float32), trans=z["trans"][:200].astype(np.float32))
print("wrote", out, os.path.getsize(out), "bytes")
