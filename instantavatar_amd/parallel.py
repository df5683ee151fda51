"""Multi-GPU helpers (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on the MI355X node, "gloo" in the CPU tests).

The hot path shards over independent units -- frames at inference (no data-path
collective), frames + ray batches at training (one gradient all-reduce per step,
instantavatar_amd.training.all_reduce_grads, plus a MAX reduction of the cached
occupancy densities every 20 steps so all ranks threshold the same field)."""
import torch


def shard_frames(n_frames, rank, world_size):
    """Round-robin frame ownership: rank r renders frames r, r+W, r+2W, ..."""
    return list(range(rank, n_frames, world_size))


def reduce_density_cache(density_cached, world_size):
    """In-place MAX all-reduce of DensityGrid.density_cached (1 MB)."""
    if world_size <= 1:
        return density_cached
    import torch.distributed as dist
    dist.all_reduce(density_cached, op=dist.ReduceOp.MAX)
    return density_cached
