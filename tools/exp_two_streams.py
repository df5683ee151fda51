"""Dev experiment: do two independent frames on two HIP streams overlap usefully?  Two model instances (own workspaces),
frames launched eagerly and alternately from one host thread; aggregate frames/s against one stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from instantavatar_amd import synthetic as syn
from instantavatar_amd.pipeline import build_synthetic_model, make_batch, GraphedRenderer

dev = torch.device("cuda", 0)
poses, tr = syn.procedural_pose_track(200)
models = [build_synthetic_model(dev, resolution=128)[0] for _ in range(2)]
res = 512
batches = [make_batch(dev, res, poses[i], tr[i]) for i in range(8)]
for m in models:
    for i in range(3):
        m.render_image_fast(batches[i], (res, res))
torch.cuda.synchronize()

def run(n_streams, n_frames=60):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_frames):
        k = i % n_streams
        with torch.cuda.stream(streams[k]):
            models[k].render_image_fast(batches[i % 8], (res, res))
    torch.cuda.synchronize()
    return n_frames / (time.perf_counter() - t0)

for rep in range(2):
    print("eager 1 stream  %.1f frames/s" % run(1))
    print("eager 2 streams %.1f frames/s" % run(2))

# graph variant: two graphs replayed on two streams
graphs = [GraphedRenderer(models[k], batches[k], (res, res), margin=1, probe_batches=[batches[j] for j in range(8)]) for k in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
def run_graph(n_streams, n_frames=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_frames):
        k = i % n_streams
        with torch.cuda.stream(streams[k]):
            graphs[k](batches[i % 8])
    torch.cuda.synchronize()
    return n_frames / (time.perf_counter() - t0)
for rep in range(2):
    print("graph 1 stream  %.1f frames/s" % run_graph(1))
    print("graph 2 streams %.1f frames/s" % run_graph(2))
