"""Microbenchmark of the field kernels in isolation (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantavatar_amd.pipeline import build_synthetic_model
dev = "cuda:0"
model, body, fp = build_synthetic_model(dev, resolution=32)
net = model.net_coarse
bb = model.deformer.bbox
g = torch.Generator(device=dev).manual_seed(0)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us
for mode in ("uniform", "sorted"):
    for V in (1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22):
        x = torch.rand((V, 3), device=dev, generator=g) * (bb[1] - bb[0]) + bb[0]
        if mode == "sorted":  # spatially coherent order (Morton-ish: sort by coarse cell)
            c = ((x - bb[0]) / (bb[1] - bb[0]) * 32).long().clamp(0, 31)
            key = (c[:, 2] * 32 + c[:, 1]) * 32 + c[:, 0]
            x = x[key.argsort()].contiguous()
        with torch.no_grad():
            t_f = timeit(lambda: net(x, None))
            t_h = timeit(lambda: net.encode(x))
            t_p = timeit(lambda: net.encode_planes(x))
            net.max_encode_workspace_bytes = 0; net._enc_ws_samples = 0; net._desc = None
            t_f0 = timeit(lambda: net(x, None))
            net.max_encode_workspace_bytes = 1 << 30; net._desc = None
        print("%-8s V=%8d  field sharded %8.1f us %5.2f Gs/s | fused %8.1f us %5.2f Gs/s || encode sharded %8.1f us %5.2f Gs/s (%4.0f GB/s) | fused %8.1f us %5.2f Gs/s" % (
            mode, V, t_f, V / t_f * 1e-3, t_f0, V / t_f0 * 1e-3, t_p, V / t_p * 1e-3, V / t_p * 1e-3 * 512, t_h, V / t_h * 1e-3))
