"""How a driver finds its place in a multi-GPU job: one process per GPU, started by `python -m torch.distributed.run
--nproc-per-node N -m instantavatar_amd.drivers.<driver> ...` (or alone: one rank, no process group).

The reference's entry points are single-GPU (`pl.Trainer(gpus=1)`, train.py:29-30; animate.py:90 `model.cuda()`); the
partitioning is SURVEY.md 8(e) / DESIGN.md section 6: frames round-robin over the ranks at inference (no data-path
collective), one frame + ray batch per rank and step at training with the gradient average over RCCL (backend "nccl" IS
RCCL on ROCm; xGMI underneath).

    launch = Launch.from_env()          # RANK / LOCAL_RANK / WORLD_SIZE (torch.distributed.run exports them)
    ... launch.device, launch.rank, launch.world_size, launch.is_main ...
    launch.close()

`IA_SHARE_DEVICE=1` (development and tests on a one-GPU box): every rank uses cuda:0 and the ranks talk over gloo -- RCCL
refuses two ranks on one device.  The control flow (sharding, gathers, broadcast, gradient average) is the N-rank one with
the real kernels; throughput figures of such a run mean nothing.
"""
import os

import torch


class Launch:
    def __init__(self, rank=0, world_size=1, local_rank=0, device=None, backend=None, shared_device=False):
        self.rank, self.world_size, self.local_rank = int(rank), int(world_size), int(local_rank)
        self.device, self.backend, self.shared_device = device, backend, bool(shared_device)
        self._own_group = False

    @property
    def is_main(self):
        return self.rank == 0

    @classmethod
    def from_env(cls, need_gpu=True, who="driver"):
        """Device from LOCAL_RANK, process group only when WORLD_SIZE > 1.  need_gpu=False (the CPU tests of the drivers'
        rank logic): no device is touched and the group is gloo."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        share = os.environ.get("IA_SHARE_DEVICE", os.environ.get("IA_BENCH_SHARE_DEVICE", "0")) == "1" and world > 1
        device = None
        if need_gpu:
            if not torch.cuda.is_available():
                raise SystemExit("%s: needs a GPU (the product path has no CPU fallback)" % who)
            if share:
                local = 0
                os.environ["IA_GRAPH_COLLECTIVES"] = "0"   # gloo collectives cannot be captured into a HIP graph
            if torch.cuda.device_count() <= local:
                raise SystemExit("%s: rank %d needs GPU %d, %d visible" % (who, rank, local, torch.cuda.device_count()))
            torch.cuda.set_device(local)
            device = torch.device("cuda", local)
        backend = None
        self = cls(rank, world, local, device, None, share)
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: the only mode this driver stack supports
            if not dist.is_initialized():
                backend = "nccl" if (need_gpu and not share) else "gloo"
                if backend == "nccl":
                    dist.init_process_group("nccl", device_id=device)
                else:
                    dist.init_process_group("gloo")
                self._own_group = True
            self.backend = dist.get_backend()
            assert dist.get_world_size() == world and dist.get_rank() == rank, "process group does not match the environment"
        return self

    # -- the few collectives a driver needs (no-ops on one rank) -------------------------------------------------
    def _comm_tensor(self, t):
        """a tensor the group's backend can move: on the device for RCCL, on the host for gloo"""
        return t.to(self.device) if self.backend == "nccl" else t.cpu()

    def barrier(self):
        if self.world_size > 1:
            import torch.distributed as dist
            dist.barrier()

    def max_over_ranks(self, x):
        if self.world_size == 1:
            return float(x)
        import torch.distributed as dist
        t = self._comm_tensor(torch.tensor([float(x)], dtype=torch.float64))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        if self.world_size == 1:
            return float(x)
        import torch.distributed as dist
        t = self._comm_tensor(torch.tensor([float(x)], dtype=torch.float64))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def gather_to_main(self, t):
        """`t` ([n, ...], same trailing shape and dtype on every rank, n may differ by rank) -> list of the ranks' tensors on
        rank 0 (host tensors), None elsewhere.  Blocks are padded to the longest for the collective."""
        if self.world_size == 1:
            return [t.cpu()]
        import torch.distributed as dist
        n = torch.tensor([t.shape[0]], dtype=torch.int64)
        ns = [torch.zeros_like(n) for _ in range(self.world_size)]
        n_c = self._comm_tensor(n)
        ns_c = [self._comm_tensor(x) for x in ns]
        dist.all_gather(ns_c, n_c)
        ns = [int(x.item()) for x in ns_c]
        pad = max(ns)
        mine = self._comm_tensor(t)
        if mine.shape[0] < pad:
            mine = torch.cat([mine, mine.new_zeros((pad - mine.shape[0],) + tuple(mine.shape[1:]))])
        mine = mine.contiguous()
        parts = [torch.empty_like(mine) for _ in range(self.world_size)] if self.is_main else None
        dist.gather(mine, parts, dst=0)
        if not self.is_main:
            return None
        return [p[:k].cpu() for p, k in zip(parts, ns)]

    def close(self):
        if self._own_group:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
            self._own_group = False
