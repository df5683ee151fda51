"""Dev tool (round 6): isolated XCD-sharded encoder time on 2^20 random points and on one frame's canonical samples, plus a
checksum of the features (cache-policy variants must be bit-identical)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from instantavatar_amd import synthetic as syn
from instantavatar_amd.pipeline import build_synthetic_model, make_batch
dev = torch.device("cuda:0")
model, body, fp = build_synthetic_model(dev)
net, bb = model.net_coarse, model.deformer.bbox
g = torch.Generator(device=dev).manual_seed(7)
n = 1 << 20
x = torch.rand((n, 3), device=dev, generator=g) * (bb[1] - bb[0]) + bb[0]
poses, tr = syn.load_animation_track(os.path.join(bench.ROOT, "tests", "golden", "aist_demo_200.npz"))
xc = bench.frame_coherent_samples(model, make_batch(dev, 512, poses[0], tr[0]), 512)
for what, pts in (("random 2^20", x), ("frame-coherent %d" % xc.shape[0], xc)):
    net.sample_coherence(not what.startswith("random"))     # (ia_field.enc_split: 2 for the random points, 3 for a frame's samples)
    us = min(bench._time_encode(net, pts, 30) for _ in range(3))
    with torch.no_grad():
        f = net.encode_planes(pts)
    f = f[0] if isinstance(f, (tuple, list)) else f
    h = hashlib.sha1(f.contiguous().cpu().numpy().tobytes()).hexdigest()[:12]
    print("%-26s %8.1f us  %.3f Gsamples/s  frac of HBM %.3f  features sha1 %s" % (what, us, pts.shape[0] / us * 1e-3, pts.shape[0] * 512 / (us * 1e-6) / 8e12, h))
