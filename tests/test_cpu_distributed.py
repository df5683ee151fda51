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


def test_single_process_is_identity():
    from instantavatar_amd.parallel import shard_frames, reduce_density_cache
    assert shard_frames(5, 0, 1) == [0, 1, 2, 3, 4]
    d = torch.ones(2, 2, 2)
    reduce_density_cache(d, 1)
    assert (d == 1).all()
