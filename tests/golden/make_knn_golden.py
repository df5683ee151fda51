"""Freezes outputs of the REFERENCE's pytorch3d CPU K-NN (third_parties/pytorch3d/cuda/knn_cpu.cpp, compiled
unmodified into oracle/_ref/ref_knn.so by oracle/build_ref.py) on the inputs of
tests/test_cpu_oracle.py::_knn_inputs -> tests/golden/knn_golden.npz (first 256 query points, K = 30).
Run in the build container (needs /root/reference):  python tests/golden/make_knn_golden.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import build_ref  # noqa: E402
from test_cpu_oracle import _knn_inputs  # noqa: E402

build_ref.build()
ref = build_ref.load_ext("ref_knn")
pts, verts = _knn_inputs()
idx, dist = ref.knn_points_idx_cpu(torch.from_numpy(pts)[None], torch.from_numpy(verts)[None], torch.tensor([len(pts)]),
                                   torch.tensor([len(verts)]), 2, 30)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "knn_golden.npz")
np.savez_compressed(out, idx=idx[0, :256].numpy(), dist=dist[0, :256].numpy())
print(out, idx.shape)
