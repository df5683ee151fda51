"""Dev tool: a few isolated encode launches (fused-all-levels vs XCD-sharded) for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantavatar_amd.pipeline import build_synthetic_model
dev = "cuda:0"
model, body, fp = build_synthetic_model(dev, resolution=32)
net = model.net_coarse
bb = model.deformer.bbox
g = torch.Generator(device=dev).manual_seed(0)
V = 1 << 20
x = torch.rand((V, 3), device=dev, generator=g) * (bb[1] - bb[0]) + bb[0]
for _ in range(4):
    net.encode(x)
    net.encode_planes(x)
torch.cuda.synchronize()
