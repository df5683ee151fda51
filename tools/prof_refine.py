"""Dev tool: the refine workload of bench.py (train.refine) alone, for rocprofv3 --kernel-trace --stats."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from instantavatar_amd import synthetic as syn
from instantavatar_amd.pipeline import build_synthetic_model
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
dev = torch.device("cuda", 0)
model, body, fp = build_synthetic_model(dev, resolution=128, n_levels=16)
poses, tr = syn.load_animation_track(os.path.join(ROOT, "tests", "golden", "aist_demo_200.npz"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
plain = "--plain" in sys.argv   # the config-2 workload (PatchSampler, fresh field) instead of refinement
r = bench.train_throughput(model, dev, poses, tr, 0, 1, n, res=512, sampler="patch" if plain else "edge", refine=not plain,
                           graphed="--eager" not in sys.argv)
print({k: r.get(k) for k in ("it_per_sec", "launch_mode", "samples_candidates_last_step", "mse_first", "mse_last", "train_overflow", "smpl_tables_max_abs_change", "graph_capture_error")})
