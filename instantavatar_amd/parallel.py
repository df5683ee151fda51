"""Multi-GPU helpers (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on the MI355X node, "gloo" in the CPU tests).

The hot path shards over independent units -- frames at inference (no data-path
collective), frames + ray batches at training:

* start / resume: `broadcast_module_state` makes every replica bit-identical to rank 0
  (parameters AND buffers: the cached occupancy densities are state too);
* every step: ONE gradient average.  The 52 MB hash-table gradient is produced last in the
  backward pass, level group by level group (`ia_hashgrid_bwd_levels`); `GradReducer` starts the
  all-reduce of a finished group on RCCL's stream while the next group is still being scattered,
  so only the last bucket's transfer is exposed.  xGMI is point-to-point (7 links per GPU), so a
  few multi-megabyte buckets beat many small ones: 4 buckets of ~6-16 MB;
* every 20 steps: MAX reduction of the cached occupancy densities (1 MB) between the EMA update
  and the thresholding, so that all ranks march the same grid.
"""
import torch


#: test / measurement switch: take the multi-rank code paths (bucketed all-reduce from inside the backward, MAX-reduce of
#: the density cache, broadcast) with a process group of ANY size, including one rank -- so that a single-GPU box
#: executes the RCCL collectives, their stream / event ordering against the scatter kernels and their graph capture
FORCE_COLLECTIVES = False


def collectives_on(world_size):
    return world_size > 1 or (FORCE_COLLECTIVES and _dist().is_available() and _dist().is_initialized())


def shard_frames(n_frames, rank, world_size):
    """Round-robin frame ownership: rank r renders frames r, r+W, r+2W, ..."""
    return list(range(rank, n_frames, world_size))


def shard_rows(n_rows, rank, world_size):
    """Intra-frame sharding for latency (optional): rank r renders the contiguous block of image rows
    [r0, r1) -- blocks differ by at most one row, empty when there are more ranks than rows."""
    base, extra = divmod(int(n_rows), int(world_size))
    r0 = rank * base + min(rank, extra)
    return r0, r0 + base + (1 if rank < extra else 0)


def _dist():
    import torch.distributed as dist
    return dist


def render_frame_tiled(model, batch, img_size, world_size=1, rank=0, jitter=None, gather=True):
    """ONE frame split over the ranks by image rows (config 5's 1024^2 frame is 7 ms on one GPU): every rank prepares the
    deformer and rebuilds the per-frame occupancy grid itself (32 KB packed, deterministic given the same `jitter`: no
    exchange), renders its block of rows through `model.render_image_fast`, and the blocks are all-gathered (RCCL; the only
    collective, 4 x 4 bytes per ray + the sample counters).  A ray's march and compositing do not depend on which other rays
    are alive -- only the per-ray SAMPLE COUNTER does (it follows the N_step schedule, raymarcher_acc.py:104, which depends
    on the number of alive rays) -- so rgb / depth / alpha equal the single-GPU frame bit for bit (tests/test_gpu_fullconfig.py).
    `jitter` must be the same on every rank: pass the tensor, or leave it None and rank 0's draw is broadcast.
    Returns (rgb, depth, alpha, counter) of the whole frame on every rank (gather=False: of this rank's rows only); the
    counter is float32 on every rank, like the renderer's."""
    import torch
    H, W = int(img_size[0]), int(img_size[1])
    r0, r1 = shard_rows(H, rank, world_size)
    sub = dict(batch)
    for k in ("rays_o", "rays_d", "near", "far"):
        sub[k] = batch[k][:, r0 * W:r1 * W].contiguous()
    if collectives_on(world_size) and world_size > 1:
        dist = _dist()
        if jitter is None:
            # every rank rebuilds the occupancy grid itself: with a private draw per rank the blocks would come from DIFFERENT
            # grids and the gathered frame would be a patchwork, silently.  One draw, rank 0's, for all (3.9 MB, once per frame)
            G = int(getattr(getattr(getattr(model, "renderer", None), "density_grid_test", None), "grid_size", 64))   # (the grid's own size, not a literal)
            jitter = torch.rand((5, G ** 3, 3), device=batch["rays_o"].device)                       # 5 = DensityGrid.initialize's default iters
            dist.broadcast(jitter, src=0)
    outs = model.render_image_fast(sub, (r1 - r0, W), jitter=jitter) if r1 > r0 else None
    if not gather or not collectives_on(world_size):
        return outs
    dist = _dist()
    dev = batch["rays_o"].device
    full = []
    shapes = [(3,), (), (), ()]
    # one fixed dtype per output on EVERY rank -- the renderer's counter is float32 (raymarcher_acc.py:185); a rank with an empty
    # row block (more ranks than rows) must not guess another one, or the all_gather mismatches
    dtypes = [torch.float32, torch.float32, torch.float32, torch.float32]
    rows = [shard_rows(H, r, world_size) for r in range(world_size)]
    pad = max(b - a for a, b in rows)          # the collective wants equal blocks: pad to the largest (they differ by <= 1 row)
    for i, (sh, dt) in enumerate(zip(shapes, dtypes)):
        parts = [torch.empty((1, pad, W) + sh, device=dev, dtype=dt) for _ in range(world_size)]
        mine = torch.zeros((1, pad, W) + sh, device=dev, dtype=dt)
        if outs is not None:
            mine[:, :r1 - r0] = outs[i].reshape((1, r1 - r0, W) + sh).to(dt)
        dist.all_gather(parts, mine)
        full.append(torch.cat([p[:, :b - a] for p, (a, b) in zip(parts, rows)], dim=1))
    return tuple(full)


def reduce_density_cache(density_cached, world_size):
    """In-place MAX all-reduce of DensityGrid.density_cached (1 MB)."""
    if not collectives_on(world_size):
        return density_cached
    dist = _dist()
    dist.all_reduce(density_cached, op=dist.ReduceOp.MAX)
    return density_cached


def broadcast_module_state(module, world_size, src=0):
    """Start-up / resume broadcast: parameters and buffers of `module` from rank `src`.
    Replicas must not rely on equal seeds: a resumed checkpoint, a different library version or a
    rank-dependent RNG draw would silently fork them."""
    if not collectives_on(world_size):
        return
    dist = _dist()
    # the training occupancy grids are kept in a plain list like the reference's (not in state_dict): name them explicitly
    extra = []
    for m in (module.modules() if hasattr(module, "modules") else []):
        for g in m.__dict__.get("density_grid_train_all", []):
            extra += list(g.buffers())
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()) + extra:
            if t.dtype == torch.bool:  # NCCL has no bool: go through uint8
                u = t.to(torch.uint8)
                dist.broadcast(u, src=src)
                t.copy_(u.bool())
            else:
                dist.broadcast(t.data, src=src)
    if hasattr(module, "modules"):
        grids = [g for m in module.modules() for g in m.__dict__.get("density_grid_train_all", [])]
        for m in list(module.modules()) + grids:
            if hasattr(m, "mark_updated"):
                m.mark_updated()       # fp16 shadows / MFMA fragments follow the new master weights
            if hasattr(m, "pack_bits") and getattr(m, "density_field", None) is not None and m.density_field.is_cuda:
                m.pack_bits()          # bit-packed occupancy follows density_field


#: set to a list to collect, over all steps, the event pairs GradReducer.finish records around its waits (see there)
EXPOSED_EVENTS = None


class GradReducer:
    """Gradient averaging with the transfers started from inside the backward pass.

    `reduce_async(t)` is called by the producer of a gradient bucket right after the kernel that
    finishes it has been enqueued: torch's process group makes the collective wait (event) for the
    work enqueued so far on the current stream and runs it on the communication stream, so later
    backward kernels overlap with the transfer.  `finish(params)` reduces whatever was not handed
    in, waits for everything and turns sums into means."""

    def __init__(self, world_size):
        self.world_size = world_size
        self._works = []
        self._ranges = []   # (first byte, last byte) already handed in
        self._fields = 0    # field calls recorded in this step's autograd graph, not yet back-propagated
        #: when set to a list, finish() appends one (event, event) pair per step around its waits on the collectives: the
        #: time the COMPUTE stream stands still for transfers that the backward did not hide (bench.py --gpus N reports it).
        #: Only meaningful for eagerly launched steps (events recorded during a graph capture cannot be timed).
        self.exposed_events = EXPOSED_EVENTS

    @property
    def active(self):
        return collectives_on(self.world_size)

    def field_forward(self):
        self._fields += 1

    def field_backward_done(self):
        """called once per field backward; True for the LAST one of the step: from then on the table gradient
        receives no further contribution and finished slices may be sent"""
        self._fields -= 1
        return self._fields == 0

    def _op(self):
        dist = _dist()
        # RCCL averages in the collective; gloo (CPU tests) has no AVG: sum now, scale in finish()
        if dist.get_backend() == "nccl":
            return dist.ReduceOp.AVG, False
        return dist.ReduceOp.SUM, True

    def reduce_async(self, t):
        if not self.active:
            return
        assert t.is_contiguous()
        op, scale = self._op()
        w = _dist().all_reduce(t, op=op, async_op=True)
        self._works.append((w, t, scale))
        b0 = t.data_ptr()
        self._ranges.append((b0, b0 + t.numel() * t.element_size()))

    def _covered(self, t):
        """bytes of `t` already handed in, as a list of (start, end) element ranges still missing"""
        b0, b1, es = t.data_ptr(), t.data_ptr() + t.numel() * t.element_size(), t.element_size()
        segs = sorted((max(a, b0), min(b, b1)) for a, b in self._ranges if a < b1 and b > b0)
        missing, cur = [], b0
        for a, b in segs:
            if a > cur:
                missing.append(((cur - b0) // es, (a - b0) // es))
            cur = max(cur, b)
        if cur < b1:
            missing.append(((cur - b0) // es, (b1 - b0) // es))
        return missing

    def finish(self, params):
        if not self.active:
            return
        for p in params:
            if p.grad is None and p.requires_grad:
                p.grad = torch.zeros_like(p)   # every rank must issue the same collectives
            g = p.grad
            if g is None:
                continue
            flat = g.view(-1) if g.is_contiguous() else None
            if flat is None:
                g = p.grad = g.contiguous()
                flat = g.view(-1)
            for a, b in self._covered(flat):
                self.reduce_async(flat[a:b])
        # every byte of every gradient travelled exactly once: the slices handed in from inside the backward
        # (training.gradient_buckets) and the remainders above must not overlap -- an overlap would average a slice twice
        segs = sorted(self._ranges)
        for (a0, b0), (a1, b1) in zip(segs, segs[1:]):
            assert a1 >= b0, "GradReducer: gradient slices overlap (%d..%d and %d..%d)" % (a0, b0, a1, b1)
        for p in params:
            if p.grad is not None:
                assert not self._covered(p.grad.view(-1)), "GradReducer: part of a gradient was not reduced"
        self.last_collectives = len(self._works)
        timed = self.exposed_events is not None and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing()
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for w, t, scale in self._works:
            w.wait()
            if scale:
                t.div_(self.world_size)
        if timed:
            e1.record()
            self.exposed_events.append((e0, e1))
        self._works, self._ranges, self._fields = [], [], 0


def exposed_allreduce_ms(pairs):
    """mean and max over the steps of the time the compute stream waited for gradient transfers (after a synchronise)"""
    ms = [a.elapsed_time(b) for a, b in pairs]
    return (sum(ms) / len(ms), max(ms)) if ms else (None, None)


#: the reducer of the training step in flight (set by training.training_step, read by the autograd
#: functions that produce gradient buckets)
_current = [None]


def current_reducer():
    return _current[0]


def set_current_reducer(r):
    _current[0] = r
