"""world_size-2 gloo tests (CPU) of the N>1 code paths: gradient averaging, the MAX
reduction of the cached density, and the frame sharding used by bench.py."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from instantavatar_amd.training import all_reduce_grads
    from instantavatar_amd.parallel import shard_frames, reduce_density_cache
    torch.manual_seed(0)
    model = torch.nn.Linear(4, 3)           # identical replicas
    x = torch.full((2, 4), float(rank + 1))  # different data per rank
    model(x).sum().backward()
    local = [p.grad.clone() for p in model.parameters()]
    all_reduce_grads(model, world)
    # expected: mean over ranks of the local gradients (rank r has grad proportional to r+1)
    exp_w = torch.full((3, 4), 2.0 * (1 + 2) / 2)
    ok = torch.allclose(model.weight.grad, exp_w) and torch.allclose(model.bias.grad, torch.full((3,), 2.0))
    dens = torch.zeros(4, 4, 4)
    dens[rank, 0, 0] = 5.0 + rank
    reduce_density_cache(dens, world)
    ok = ok and dens[0, 0, 0] == 5.0 and dens[1, 0, 0] == 6.0
    frames = shard_frames(7, rank, world)
    q.put((rank, bool(ok), frames, [float(g.sum()) for g in local]))
    dist.destroy_process_group()


def test_gloo_world2_grad_average_density_max_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
    f0, f1 = res[0][2], res[1][2]
    assert sorted(f0 + f1) == list(range(7)) and not set(f0) & set(f1)  # disjoint cover
    assert f0 == [0, 2, 4, 6] and f1 == [1, 3, 5]


def _tile_worker(rank, world, port, q):
    import torch.distributed as dist
    from instantavatar_amd.parallel import render_frame_tiled, shard_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, W = 7, 5                      # 7 rows over 2 ranks: blocks of 4 and 3 rows

    class Fake:                      # stands in for AvatarModel: "renders" a function of the ray it is given
        jitters = []

        def render_image_fast(self, batch, img_size, jitter=None):
            o = batch["rays_o"]
            assert o.shape[1] == img_size[0] * img_size[1]
            rgb = o.reshape(1, *img_size, 3) * 2.0
            dep = batch["near"].reshape(1, *img_size) + 1.0
            self.jitters.append(None if jitter is None else float(jitter.sum()))
            return rgb, dep, dep * 0.5, dep * 10          # the renderer's counter is float32 (raymarcher_acc.py:185)
    idx = torch.arange(H * W, dtype=torch.float32)
    batch = {"rays_o": torch.stack([idx, idx + 0.25, idx + 0.5], -1)[None], "rays_d": torch.zeros(1, H * W, 3),
             "near": idx[None].clone(), "far": idx[None] + 2}
    rgb, dep, alpha, cnt = render_frame_tiled(Fake(), batch, (H, W), world, rank)
    ok = (rgb.shape == (1, H, W, 3) and torch.equal(rgb.reshape(-1, 3), batch["rays_o"][0] * 2) and torch.equal(dep.reshape(-1), idx + 1)
          and torch.equal(alpha.reshape(-1), (idx + 1) * 0.5) and cnt.dtype == torch.float32 and torch.equal(cnt.reshape(-1), (idx + 1) * 10))
    # no jitter given: rank 0's draw reaches every rank (private draws would give every block its own occupancy grid)
    js = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(js, torch.tensor([Fake.jitters[-1]], dtype=torch.float64))
    ok = ok and Fake.jitters[-1] is not None and all(float(j) == float(js[0]) for j in js)
    # more ranks than rows: the rank with the empty block joins the gather with the same dtypes (it used to guess int32)
    b1 = {k: v[:, :W].clone() for k, v in batch.items()}
    rgb1, dep1, alpha1, cnt1 = render_frame_tiled(Fake(), b1, (1, W), world, rank)
    ok = ok and rgb1.shape == (1, 1, W, 3) and cnt1.dtype == torch.float32 and torch.equal(cnt1.reshape(-1), (idx[:W] + 1) * 10)
    q.put((rank, bool(ok), shard_rows(H, rank, world)))
    dist.destroy_process_group()


def test_gloo_world2_intra_frame_row_sharding_gathers_the_whole_frame():
    """one frame split by image rows over the ranks and all-gathered; ragged blocks (7 rows on 2 ranks)."""
    from instantavatar_amd.parallel import shard_rows
    assert [shard_rows(7, r, 2) for r in range(2)] == [(0, 4), (4, 7)]
    assert [shard_rows(2, r, 3) for r in range(3)] == [(0, 1), (1, 2), (2, 2)]          # more ranks than rows: an empty block
    assert [shard_rows(1024, r, 8) for r in (0, 7)] == [(0, 128), (896, 1024)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tile_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res) and [r[2] for r in res] == [(0, 4), (4, 7)]


def test_single_process_is_identity():
    from instantavatar_amd.parallel import shard_frames, reduce_density_cache
    assert shard_frames(5, 0, 1) == [0, 1, 2, 3, 4]
    d = torch.ones(2, 2, 2)
    reduce_density_cache(d, 1)
    assert (d == 1).all()


# ---------------------------------------------------------------------------------------------
# bench.py --gpus 2: self-launch through torch.distributed.run, frame sharding, collectives (gloo, no kernels)
# ---------------------------------------------------------------------------------------------
def test_bench_gpus2_dry_run_launches_two_ranks():
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--dry-run"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and r["dry_run"] is True and r["frames_per_rank"] == [6, 6] and r["steps"] == 6
    assert r["metric"] == "novel_pose_render_frames_per_sec_512x512" and r["scaling"] == "weak"
    # every rank reports what it saw: gathered into the line AND echoed on stderr by each process
    assert [x["rank"] for x in r["ranks"]] == [0, 1] and all(x["world_size_seen"] == 2 and x["backend"] == "gloo" for x in r["ranks"])
    assert len({x["pid"] for x in r["ranks"]}) == 2
    assert "[bench rank 0/2]" in out.stderr and "[bench rank 1/2]" in out.stderr
    # a mismatch between --gpus and the launcher's world size must fail loudly
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--dry-run"],
                         env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port())),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)


def _bench_dry(extra_env, extra_args=()):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run", "--train-only"] + list(extra_args),
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_supervised_training_child_job_dry_run():
    """bench.supervised_train without kernels (VERDICT r04 task 7): with N > 1 the training phase runs in a child job -- each
    rank re-executes bench.py as its own child on the next rendezvous port (WITHOUT torchrun's TORCHELASTIC_* variables: the
    children must host their own store), rank 0 parses its child's line.  Then the watchdog's case: rank 1's first child never
    gets anywhere; both parents give up after IA_BENCH_CHILD_TIMEOUT, kill their children by PID, agree on the failure and run
    the phase again with IA_GRAPH_COLLECTIVES=0 -- the line reports both attempts."""
    r = _bench_dry({})
    sup = r["train"]["supervised"]["attempts"]
    assert len(sup) == 1 and sup[0]["all_ranks_ok"] and sup[0]["IA_GRAPH_COLLECTIVES"] == "1", sup
    assert r["train"]["ranks_in_child_group"] == 2 and r["train"]["graph_collectives"] is True and r["value"] == 8192.0
    r = _bench_dry({"IA_TEST_CHILD_HANG_RANK": "1", "IA_BENCH_CHILD_TIMEOUT": "30"})
    sup = r["train"]["supervised"]["attempts"]
    assert len(sup) == 2 and not sup[0]["all_ranks_ok"] and "killed" in sup[0]["this_rank"], sup
    assert sup[1]["all_ranks_ok"] and sup[1]["IA_GRAPH_COLLECTIVES"] == "0" and r["train"]["graph_collectives"] is False, sup


def test_bench_gpus_without_devices_fails_loudly():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a box with fewer than 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "GPU(s) visible" in (out.stderr + out.stdout)


# ---------------------------------------------------------------------------------------------
# the REAL training.training_step on two gloo ranks, with the kernels' outputs mocked:
# start-up broadcast, bucketed gradient all-reduce started inside backward, density MAX-reduce hook,
# non-finite-gradient skip -- replicas must stay bit-identical although every rank sees different data.
# ---------------------------------------------------------------------------------------------
class _MockParams(torch.nn.Module):
    def __init__(self, n):
        super().__init__()
        self.params = torch.nn.Parameter(torch.randn(n))


class _MockNet(torch.nn.Module):
    """NeRFNGPNet's surface as training_step uses it: encoder.params / color_net.params, initialize, mark_updated."""

    def __init__(self):
        super().__init__()
        self.encoder = _MockParams(96)
        self.color_net = _MockParams(24)
        self.updates = 0

    def initialize(self, bbox):
        pass

    def mark_updated(self):
        self.updates += 1


class _BucketFieldFn(torch.autograd.Function):
    """stands in for training._FieldFn: accumulates into .grad in place and hands finished buckets to the reducer
    from inside backward, exactly like the hash-grid backward does with its level groups"""

    @staticmethod
    def forward(ctx, x, enc_params, col_params, net):
        from instantavatar_amd import parallel
        red = parallel.current_reducer()
        if red is not None:
            red.field_forward()
        ctx.net = net
        ctx.save_for_backward(x)
        e, c = net.encoder.params.detach(), net.color_net.params.detach()
        return (x * e[:3]).sum(-1) + c.sum() * 0.01

    @staticmethod
    def backward(ctx, g):
        from instantavatar_amd import parallel
        from instantavatar_amd.training import _grad_buffer
        (x,) = ctx.saved_tensors
        net = ctx.net
        ge, gc = _grad_buffer(net.encoder.params), _grad_buffer(net.color_net.params)
        gc += g.sum() * 0.01
        red = parallel.current_reducer()
        last = red is not None and red.active and red.field_backward_done()
        ge[:3] += (g[:, None] * x).sum(0)
        ge[48:] += g.mean()
        if last:
            red.reduce_async(gc)
            red.reduce_async(ge[48:])   # "fine levels" first ...
            red.reduce_async(ge[:48])   # ... the bucket with the MLP weights last
        return None, None, None, None


class _MockDeformer:
    def __init__(self):
        self.bbox = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
        self.initialized = True

    def prepare_deformer(self, batch):
        pass

    def transform_rays_w2s(self, rays):
        pass

    def __call__(self, pts, net, eval_mode=True):
        sigma = _BucketFieldFn.apply(pts, net.encoder.params, net.color_net.params, net)
        return torch.zeros_like(pts), sigma


def _train_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from instantavatar_amd import parallel, training
    from instantavatar_amd.models.structures.density_grid import DensityGrid

    torch.manual_seed(1000 + rank)             # different replicas AND different data per rank
    net = _MockNet()

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net_coarse = net
            self.deformer = _MockDeformer()
            self.global_step = 0
            grid = DensityGrid(4, aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]))
            # the occupancy post-process is a HIP kernel: mocked by a plain threshold of the (reduced) cache
            grid._postprocess = lambda density: setattr(grid, "density_field", density > density.mean())
            self.renderer = type("R", (), {"density_grid_train": grid, "idx": 0})()

        def forward(self, batch, eval_mode=False, noise=0):
            _, sigma = self.deformer(batch["pts"], self.net_coarse, eval_mode=False)
            a = torch.sigmoid(sigma)
            return {"rgb_coarse": a[:, None].expand(-1, 3), "alpha_coarse": a, "weight_coarse": a[:, None].expand(-1, 4)}

    model = Model()
    parallel.broadcast_module_state(model, world)      # start-up broadcast: replicas become identical
    p0 = [p.detach().clone() for p in model.parameters()]
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    loss_fn = training.NeRFLoss(fused=False)
    ok = True
    for step in range(22):                               # steps 0 and 20 run the density update (two field calls)
        batch = {"pts": torch.randn(16, 3), "rgb": torch.rand(16, 3), "alpha": torch.rand(16)}
        if step == 5:                                    # local gradients at the CURRENT parameters, no reducer
            opt.zero_grad(set_to_none=True)
            loss_fn(model.forward(batch), batch)["loss"].backward()
            g_loc = torch.cat([model.net_coarse.encoder.params.grad, model.net_coarse.color_net.params.grad]).clone()
            both = [torch.zeros_like(g_loc) for _ in range(world)]
            dist.all_gather(both, g_loc)
        losses = training.training_step(model, batch, opt, loss_fn, world_size=world)
        if step == 5:                                    # what the step reduced = mean over ranks of the local gradients
            g_avg = torch.cat([model.net_coarse.encoder.params.grad, model.net_coarse.color_net.params.grad])
            ok = ok and torch.allclose(g_avg, sum(both) / world, atol=1e-6) and not torch.allclose(both[0], both[1], atol=1e-4)
    # a non-finite gradient on ONE rank: every rank must skip the step, parameters stay finite and identical
    before = [p.detach().clone() for p in model.parameters()]
    batch = {"pts": torch.randn(16, 3), "rgb": torch.rand(16, 3), "alpha": torch.rand(16)}
    if rank == 1:
        batch["pts"][0, 0] = float("nan")
    losses = training.training_step(model, batch, opt, loss_fn, world_size=world)
    skipped = bool(losses["skipped_non_finite"])
    same_after_skip = all(torch.equal(a, b.detach()) for a, b in zip(before, model.parameters()))
    # a candidate-capacity overflow on ONE rank (round 3): that rank's loss is turned into NaN before the backward pass, the
    # gradient average carries it to the other rank, both skip -- no extra collective, replicas stay identical
    before2 = [p.detach().clone() for p in model.parameters()]
    batch = {"pts": torch.randn(16, 3), "rgb": torch.rand(16, 3), "alpha": torch.rand(16)}
    fwd = model.forward

    def forward_with_overflow(b, eval_mode=False, noise=0):
        out = fwd(b, eval_mode=eval_mode, noise=noise)
        model.renderer.train_overflow_flag = torch.tensor(1.0 if rank == 0 else 0.0)     # what render_train_fused sets
        return out
    model.forward = forward_with_overflow
    losses2 = training.training_step(model, batch, opt, loss_fn, world_size=world)
    model.forward = fwd
    skipped = skipped and bool(losses2["skipped_non_finite"]) and float(losses2["skipped_overflow"]) == (1.0 if rank == 0 else 0.0)
    same_after_skip = same_after_skip and all(torch.equal(a, b.detach()) for a, b in zip(before2, model.parameters()))
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()] + [model.renderer.density_grid_train.density_cached.reshape(-1)])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    identical = all(torch.equal(gathered[0], g) for g in gathered)
    q.put((rank, bool(ok), identical, skipped, same_after_skip, float(p0[0].sum()), model.global_step, net.updates))
    dist.destroy_process_group()


def test_gloo_world2_real_training_step_plumbing():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, identical, skipped, same_after_skip, p0sum, gstep, updates in res:
        assert ok, "averaged gradient != mean of the local gradients"
        assert identical, "replicas (parameters + cached densities) diverged"
        assert skipped and same_after_skip, "a non-finite gradient must skip the optimiser step on every rank"
        assert gstep == 24 and updates == 25   # 24 steps + the start-up broadcast
    assert res[0][5] == res[1][5], "start-up broadcast did not equalise the replicas"


def test_gradient_buckets_tile_the_encoder_gradient():
    """The slices of `encoder.params.grad` handed to the all-reduce from inside the hash-grid backward (one per level
    group, finest first) must be disjoint and cover the vector exactly, for both grid depths and both tcnn layouts."""
    from instantavatar_amd.models.networks.ngp import NeRFNGPNet
    from instantavatar_amd.training import gradient_buckets
    for n_levels in (16, 8):
        for r3 in (54, 55):
            net = NeRFNGPNet(dict(center=[0, 0, 0], scale=[1, 1, 1]), n_levels=n_levels, level3_res=r3)
            b = gradient_buckets(net)
            levels = [lv for lv, _ in b]
            assert levels == sorted(levels, reverse=True) and levels[-1][0] == 0 and levels[0][1] == n_levels
            assert all(levels[i][0] == levels[i + 1][1] for i in range(len(levels) - 1))          # contiguous level ranges
            spans = sorted(s for _, s in b)
            assert spans[0][0] == 0 and spans[-1][1] == net.encoder.params.numel()
            assert all(spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))             # disjoint, no gap
            w_end = net.sig_w1_size + 1024
            off = [int(o) for o in net.hash_desc.offset[:n_levels + 1]]
            for (l0, l1), (lo, hi) in b:                                                            # a slice holds exactly its levels (+ the MLP weights)
                assert hi == w_end + 2 * off[l1] and lo == (0 if l0 == 0 else w_end + 2 * off[l0])


# --------------------------------------------------------------------------------------------------------------------
# drivers.animate.render_sequence under a 2-rank launch (VERDICT r04 task 1a): frame sharding, the per-rank PNG files, the
# gather of the packed frames to rank 0 for the GIF and the eager re-render of frames the graph could not finish -- with the
# renderer replaced by a function of the pose (the kernels need a GPU; tests/test_gpu_drivers.py runs the real ones).
def _fake_frame(batch, H, W):
    """a frame that depends on the pose alone: rgb / alpha in [0, 1], all four channels distinct per frame"""
    s = batch["global_orient"].sum() + batch["transl"].sum()
    base = torch.linspace(0, 1, H * W).reshape(1, H, W)
    rgb = torch.stack([(base + s * 0.37) % 1.0, (base * 0.5 + s * 0.11) % 1.0, (base * 0.25 + s * 0.73) % 1.0], -1)
    alpha = (base + s * 0.05) % 1.0
    return rgb, alpha * 0, alpha, alpha * 0


def _fake_pack(out, dst):
    rgb, _, alpha, _ = out
    dst.copy_((torch.cat([rgb, alpha[..., None]], -1)[0].clamp(0, 1) * 255).to(torch.uint8))   # animate.py:107-113


class _FakePipelined:
    """the interface of pipeline.PipelinedRenderer that render_sequence uses"""

    def __init__(self, seq, unfinished):
        self.seq, self.calls, self.unfinished = seq, 0, unfinished

    def __call__(self, batch, consume=None):
        assert "rays_o" not in batch, "the frame loop hands the graphs the SMPL parameters only"
        out = _fake_frame(batch, self.seq.H, self.seq.W)
        if self.calls in self.unfinished:     # a frame whose wave-front loop ran out of captured iterations: garbage until re-rendered
            out = tuple(o * 0 + 0.5 for o in out)
        k = self.calls % 2
        consume(out, k)
        self.calls += 1
        return out, k

    def synchronize(self):
        pass

    def finish(self):
        return len(self.unfinished)

    @property
    def incomplete_calls(self):
        return sorted(self.unfinished)


def _animate_worker(rank, world, port, out_dir, q):
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    from instantavatar_amd.drivers import animate
    from instantavatar_amd.drivers.launch import Launch
    from instantavatar_amd import synthetic
    launch = Launch.from_env(need_gpu=False)
    assert launch.world_size == world and launch.rank == rank and (launch.backend == "gloo") == (world > 1)
    poses, trans = synthetic.procedural_pose_track(8)
    seq = animate.AnimateSequence(poses[:7], trans[:7], np.zeros(10, np.float32), "cpu", downscale=90)   # 12 x 12 pixels, 7 frames
    animate.pack_rgba8 = _fake_pack

    class Model:
        def render_image_fast(self, batch, size, jitter=None):
            return _fake_frame(batch, *size)
    made = []

    def make(model, first, size, in_flight, probes, jitter):
        assert "rays_o" in first and all("rays_o" in p for p in probes) and size == (seq.H, seq.W)
        made.append(_FakePipelined(seq, {1} if rank == 0 else set()))
        return made[-1]
    res = animate.render_sequence(Model(), seq, out_dir, gif="a.gif", launch=launch, make_renderer=make, log=None)
    # the same sequence without a GIF through frame buffers that hold TWO frames: rendered and written chunk by chunk
    res2 = animate.render_sequence(Model(), seq, out_dir + "_chunked", gif=None, launch=launch, make_renderer=make, log=None,
                                   max_buffer_bytes=2 * seq.H * seq.W * 4)
    assert (res2["frames"], res2["local"], res2["incomplete"]) == (res["frames"], res["local"], res["incomplete"]) and made[1].calls == made[0].calls
    q.put((rank, res["frames"], res["local"], res["incomplete"], made[0].calls))
    launch.close()


def _run_animate(world, out_dir):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_animate_worker, args=(r, world, port, out_dir, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_gloo_world2_animate_driver_shards_frames_and_gathers_the_gif(tmp_path):
    import numpy as np
    from PIL import Image
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    r1 = _run_animate(1, one)
    r2 = _run_animate(2, two)
    assert r1 == [(0, 7, 7, 1, 7)]
    assert r2 == [(0, 7, 4, 1, 4), (1, 7, 3, 0, 3)]          # round-robin: rank 0 frames 0 2 4 6, rank 1 frames 1 3 5
    for i in range(7):
        a, b = open(os.path.join(one, "%d.png" % i), "rb").read(), open(os.path.join(two, "%d.png" % i), "rb").read()
        assert a == b, "frame %d differs between the 1-rank and the 2-rank run" % i
    assert sorted(os.listdir(two)) == sorted(["%d.png" % i for i in range(7)] + ["a.gif"])
    for d in (one, two):                                     # the chunked, GIF-less pass wrote the same files
        assert sorted(os.listdir(d + "_chunked")) == sorted("%d.png" % i for i in range(7))
        for i in range(7):
            assert open(os.path.join(d + "_chunked", "%d.png" % i), "rb").read() == open(os.path.join(one, "%d.png" % i), "rb").read(), (d, i)
    frames = [np.asarray(Image.open(os.path.join(one, "%d.png" % i))) for i in range(7)]
    assert all(not np.array_equal(frames[0], f) for f in frames[1:])
    assert not (frames[2][..., 0] == 127).all(), "the unfinished frame was not rendered again"    # (call 1 of rank 0 = frame 2)
    g1, g2 = Image.open(os.path.join(one, "a.gif")), Image.open(os.path.join(two, "a.gif"))
    assert g1.n_frames == g2.n_frames == 7
    for i in range(7):                                       # the gathered frames are interleaved back into sequence order
        g1.seek(i), g2.seek(i)
        assert np.array_equal(np.asarray(g1.convert("RGBA")), np.asarray(g2.convert("RGBA"))), i


# --------------------------------------------------------------------------------------------------------------------
# GraphedTrainStep at world size 2 (VERDICT r04 weak 6 / task 7): the ranks agree on the outcome of the FIRST capture -- one rank
# whose capture failed must not launch its collectives eagerly against the other rank's replays.  The capture itself needs a
# GPU; here `_capture` is replaced by a stand-in whose "graph" replays an eager step, everything around it is the real code.
def _graph_agreement_worker(rank, world, port, fail_rank, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from instantavatar_amd import parallel, training
    from instantavatar_amd.models.structures.density_grid import DensityGrid
    torch.manual_seed(7 + rank)
    net = _MockNet()

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net_coarse, self.deformer, self.global_step = net, _MockDeformer(), 0
            grid = DensityGrid(4, aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]))
            grid._postprocess = lambda density: setattr(grid, "density_field", density > density.mean())
            self.renderer = type("R", (), {"density_grid_train": grid, "idx": 0, "train_cand_capacity": 1,
                                           "_train_counts_check": lambda self: None, "_train_counts_peek": lambda self, cap: None})()

        def forward(self, batch, eval_mode=False, noise=0):
            _, sigma = self.deformer(batch["pts"], self.net_coarse, eval_mode=False)
            a = torch.sigmoid(sigma)
            return {"rgb_coarse": a[:, None].expand(-1, 3), "alpha_coarse": a, "weight_coarse": a[:, None].expand(-1, 4)}
    model = Model()
    parallel.broadcast_module_state(model, world)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    loss_fn = training.NeRFLoss(fused=False)
    stepper = training.GraphedTrainStep(model, opt, loss_fn, world_size=world, graph_collectives=True)
    assert stepper.enabled

    def fake_capture(key, use_noise):
        if fail_rank is not None and rank == fail_rank:      # the fault is injected HERE, in the stand-in: no test hook in the product
            raise RuntimeError("capture failure injected by the test")

        class G:
            def replay(self_inner):
                entry["out"] = training.training_step(model, stepper.inputs, opt, loss_fn, world, _capturing=True)
        params = [p for g in opt.param_groups for p in g["params"]]
        entry = dict(graph=G(), out=None, grads=[p.grad for p in params], params=params, cap=1)
        stepper.graphs[key] = entry
        return entry
    stepper._capture = fake_capture
    for step in range(6):
        stepper({"pts": torch.randn(16, 3), "rgb": torch.rand(16, 3), "alpha": torch.rand(16)})
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    q.put((rank, stepper.enabled, stepper.replays, stepper.eager_steps, stepper.capture_error, torch.equal(both[0], both[1]), model.global_step))
    dist.destroy_process_group()


def _run_graph_agreement(fail_rank):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_graph_agreement_worker, args=(r, 2, port, fail_rank, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_gloo_world2_first_graph_capture_is_all_or_nothing():
    ok = _run_graph_agreement(None)
    for rank, enabled, replays, eager, err, identical, gstep in ok:
        assert enabled and replays == 5 and eager == 1 and err is None and identical and gstep == 6
    bad = _run_graph_agreement(1)                       # the capture fails on rank 1 ONLY
    for rank, enabled, replays, eager, err, identical, gstep in bad:
        assert not enabled and replays == 0 and eager == 6 and identical and gstep == 6, (rank, enabled, replays, eager, err)
    assert "injected" in bad[1][4] and "another rank" in bad[0][4]
