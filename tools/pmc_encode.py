"""Dev tool: isolated encode launches for rocprofv3 --pmc passes -- 2^20 uniformly random points (fused-all-levels kernel
and the XCD-sharded kernel) and, with `coherent` as argument, the canonical samples of a real 512^2 frame in pipeline
order (what bench.py's hashgrid_lookup.frame_coherent times)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantavatar_amd.pipeline import build_synthetic_model, make_batch
dev = "cuda:0"
coherent = len(sys.argv) > 1 and sys.argv[1].startswith("coherent")      # coherent | coherent-morton | coherent-shuffled (the same points re-ordered)
model, body, fp = build_synthetic_model(dev, resolution=128 if coherent else 32)
net = model.net_coarse
bb = model.deformer.bbox
if coherent:
    import importlib.util
    from instantavatar_amd import synthetic as syn
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    poses, tr = syn.load_animation_track(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "aist_demo_200.npz"))
    x = bench.frame_coherent_samples(model, make_batch(dev, 512, poses[0], tr[0]), 512)
    if sys.argv[1] == "coherent-morton":
        x = x[bench._morton_order(x, bb)[0]].contiguous()
    elif sys.argv[1] == "coherent-shuffled":
        x = x[torch.randperm(x.shape[0], device=dev, generator=torch.Generator(device=dev).manual_seed(1))].contiguous()
    print("frame-coherent samples (%s):" % sys.argv[1], x.shape[0])
    torch.cuda.synchronize()
    for _ in range(4):
        net.encode_planes(x)
else:
    g = torch.Generator(device=dev).manual_seed(0)
    V = 1 << 20
    x = torch.rand((V, 3), device=dev, generator=g) * (bb[1] - bb[0]) + bb[0]
    net.sample_coherence(False)     # random points: the even split of the encoder's second level group (as bench.py measures them)
    for _ in range(4):
        net.encode(x)
        net.encode_planes(x)
torch.cuda.synchronize()
