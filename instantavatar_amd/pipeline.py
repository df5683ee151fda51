"""Lightning-free caller of the three plugins: mirrors what DNeRFModel does on the
hot path (instant_avatar/models/DNeRF.py:61-97 `forward` / `render_image_fast`),
so benchmarks, smoke tests and parity tests exercise the plugins exactly the way
train.py / animate.py would, without pytorch_lightning / hydra (absent here).
"""
import numpy as np
import torch

from . import synthetic
from .deformers.smplx import SMPL
from .deformers.snarf_deformer import SNARFDeformer
from .models.networks.ngp import NeRFNGPNet
from .models.structures.utils import Rays
from .renderers.raymarcher_acc import Raymarcher


class AvatarModel(torch.nn.Module):
    """net_coarse / deformer / renderer wiring of DNeRFModel.__init__ (DNeRF.py:22-28)."""

    def __init__(self, deformer, net, renderer):
        super().__init__()
        self.net_coarse = net
        self.deformer = deformer
        self.renderer = renderer
        self.global_step = 0
        self.is_refine = False      # opt.optimize_SMPL.is_refine (confs/SNARF_NGP_refine.yaml): render with the refined SMPL tables

    def forward(self, batch, eval_mode=True, noise=0):
        """DNeRF.py:61-70"""
        rays = Rays(o=batch["rays_o"], d=batch["rays_d"], near=batch["near"], far=batch["far"])
        self.deformer.transform_rays_w2s(rays)
        return self.renderer(rays, lambda x, _: self.deformer(x, self.net_coarse, eval_mode), eval_mode=eval_mode,
                             noise=noise, bg_color=batch.get("bg_color", None))

    @torch.no_grad()
    def render_image_fast(self, batch, img_size, jitter=None):
        """DNeRF.py:72-97: (refinement: the frame's SMPL parameters come from the optimised tables, near / far follow the
        refined translation -- :73-86), prepare deformer, rebuild the test occupancy grid, render."""
        if getattr(self, "SMPL_param", None) is not None and self.is_refine:
            idx = batch["idx_dev"] if torch.is_tensor(batch.get("idx_dev")) else batch["idx"].reshape(-1).long().to(batch["transl"].device)
            body_params = self.SMPL_param(idx.reshape(-1).long())
            for k in ("global_orient", "body_pose", "transl"):
                assert batch[k].shape == body_params[k].shape, (k, batch[k].shape, body_params[k].shape)
                batch[k] = body_params[k]
            if type(self.deformer).__name__ == "SMPLDeformer":
                batch["betas"] = body_params["betas"]
            dist = torch.norm(batch["transl"], dim=-1, keepdim=True).detach()
            batch["near"][:] = dist - 1
            batch["far"][:] = dist + 1
        self.deformer.prepare_deformer(batch)
        self.renderer.density_grid_test.initialize(self.deformer, self.net_coarse, jitter=jitter)
        d = self.forward(batch, eval_mode=True)
        rgb = d["rgb_coarse"].reshape(-1, *img_size, 3)
        depth = d["depth_coarse"].reshape(-1, *img_size)
        alpha = d["alpha_coarse"].reshape(-1, *img_size)
        counter = d["counter_coarse"].reshape(-1, *img_size)
        return rgb, depth, alpha, counter


class GraphedRenderer:
    """render_image_fast replayed from a HIP graph (torch.cuda.CUDAGraph).

    The per-frame pipeline has no host synchronisation and fixed launch geometry (counts
    live on the device), so one capture serves every frame: pose / translation / near /
    far are copied into static buffers and the graph is replayed.  The wave-front loop's
    length is data dependent; the graph holds `margin` more iterations than the warm-up
    frames needed (idle iterations are a few empty launches) and the device-side alive
    counter of every frame is copied to pinned host memory and inspected a few calls later without
    blocking (`incomplete` counts frames whose loop would have continued, `incomplete_calls` lists
    them; they must be re-rendered through `model.render_image_fast`).  `sync_check=True` checks
    before returning."""

    def __init__(self, model, batch, img_size, warmup=3, margin=4, sync_check=False, probe_batches=(), jitter=None):
        """probe_batches: further batches (other poses of the sequence) rendered once eagerly to measure
        how many wave-front iterations the sequence needs; with a representative sample a small
        `margin` is enough (every idle iteration costs five empty launches per frame).
        jitter: a fixed occupancy-probe jitter ([5, 64^3, 3] in [0, 1), DensityGrid.initialize) read by every replay
        instead of a fresh draw -- frames then depend on their pose only, whichever rank / replica / replay renders them."""
        self.model, self.img_size, self.sync_check, self.jitter = model, img_size, sync_check, jitter
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        go, bp = self.static.get("global_orient"), self.static.get("body_pose")
        if torch.is_tensor(go) and torch.is_tensor(bp) and go.numel() == 3 and bp.numel() == 69 and go.dtype == bp.dtype == torch.float32:
            # the two pose inputs as the halves of ONE 72-float record: prepare_deformer hands the record to ia_smpl_tfs as it
            # is (snarf_deformer._pose72) -- no concatenation launch in the captured frame
            rec = torch.cat([go.reshape(-1), bp.reshape(-1)])
            self.static["global_orient"], self.static["body_pose"] = rec[:3].view(go.shape), rec[3:].view(bp.shape)
        r = model.renderer
        need = 0
        for b in probe_batches:
            model.render_image_fast(b, img_size, jitter=jitter)
            need = max(need, r.iters_executed())
        for _ in range(warmup):  # settles workspace sizes, fp16 shadows and the iteration count
            model.render_image_fast(self.static, img_size, jitter=jitter)
            need = max(need, r.iters_executed())
        r._iters_hint = need + margin   # (no parity constraint: the alive lists ping-pong on the absolute iteration index)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        r._graph_capture = True
        try:
            # thread_local: other threads of the process (e.g. the RCCL watchdog of a multi-GPU job)
            # may touch the runtime while this thread captures
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.out = model.render_image_fast(self.static, img_size, jitter=jitter)
        finally:
            r._graph_capture = False
        # deferred alive check: a ring of pinned slots, polled without blocking -- the host may run up
        # to DEPTH frames ahead of the GPU (blocking on the previous frame before launching the next
        # one would leave the GPU idle for the host's wake-up + enqueue time every frame)
        self.DEPTH = 4
        self._host = torch.zeros(self.DEPTH, dtype=torch.int32).pin_memory()
        self._pending = []           # [(event, slot, call index)], oldest first
        self.incomplete = 0
        self.calls = 0
        self.incomplete_calls = []   # indices (0-based call numbers) of the frames that must be re-rendered

    def _poll(self, block_all=False):
        while self._pending:
            ev, slot, call = self._pending[0]
            if not ev.query():
                if not block_all and len(self._pending) < self.DEPTH:
                    return
                ev.synchronize()
            if int(self._host[slot]) > 0:
                self.incomplete += 1
                self.incomplete_calls.append(call)
            self._pending.pop(0)

    def __call__(self, batch):
        self._poll()
        slot = self.calls % self.DEPTH
        # (near / far of the batch are dead inputs on this path: transform_rays_w2s recomputes them from the ray origins in
        # the SMPL-root frame, snarf_deformer.py:101-103 -- two 1 MB copies per frame that nobody read)
        dst = [self.static[k] for k in ("global_orient", "body_pose", "transl")]
        src = [batch[k] for k in ("global_orient", "body_pose", "transl")]
        if any(d.data_ptr() != s_.data_ptr() for d, s_ in zip(dst, src)):
            torch._foreach_copy_(dst, src, non_blocking=True)   # one launch for the 75 pose floats
        self.graph.replay()
        self._host[slot:slot + 1].copy_(self.model.renderer._n_alive_dev[:1], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pending.append((ev, slot, self.calls))
        self.calls += 1
        if self.sync_check:
            self._poll(block_all=True)
        return self.out

    def finish(self):
        self._poll(block_all=True)
        return self.incomplete


def clone_for_stream(model):
    """A second AvatarModel for the same avatar: network weights, body model and skinning-weight voxels are SHARED by
    reference, everything a frame writes (per-frame deformer state, occupancy grid, render workspaces, encoder scratch) is
    its own -- so that two frames can be in flight on two streams."""
    renderer = Raymarcher(model.renderer.MAX_SAMPLES, model.renderer.MAX_BATCH_SIZE, smpl_init=model.renderer.smpl_init)
    renderer = renderer.to(model.renderer.aabb.device)
    renderer.initialize(1)
    return AvatarModel(model.deformer.clone_shared(), model.net_coarse.clone_shared(), renderer)


class PipelinedRenderer:
    """`n_in_flight` frames in flight: one captured HIP graph per replica (GraphedRenderer), replayed on its own
    stream (round robin, or the least loaded replica: `schedule`).  A frame is a chain of ~60 dependent launches of which only the Broyden search and the encoder fill the chip;
    the marcher, the compositor, the occupancy post-process and the small first / last wave-front iterations are latency
    bound and leave most CUs idle -- other, independent frames (animate.py renders independent frames: BASELINE config 3)
    run in those gaps.  Outputs of a call stay valid until the replica that rendered it is called again (with the round-robin
    schedule: call i + n_in_flight); consumers belong into `consume`, which runs on the replica's stream right behind the frame.

    Stream priorities (round 6): with equal priorities the hardware queues share the dispatcher evenly and a third frame LOSES
    (513 vs 553 frames/s with two); with the FIRST replica's stream at high priority and the others at normal priority three
    frames in flight WIN (580 sustained / 598 over the 20-frame driver window at 512^2: `profiles/r06_ab_in_flight_priorities.txt`):
    the high-priority frame's chain of small launches is dispatched the moment it is ready instead of queueing behind the
    thousands of pending search workgroups of the other frames, which fill what it leaves free.  Two in flight gain nothing from
    it (545 vs 553), four and five lose to three; handing each frame to the least loaded replica instead of round robin adds 1 %
    (605 vs 600).  Default: (high, normal, normal, ...) from three frames in flight on, equal
    below; `priorities` (a list, HIP convention: lower = higher priority, range `torch.cuda.Stream.priority_range()`) or the
    environment variable IA_STREAM_PRIORITIES ("-1,0,0") override.  Measured on MI355X: one frame in flight 450 frames/s
    (2.2 ms latency), two 553, three 580."""

    def __init__(self, model, batch, img_size, n_in_flight=3, margin=1, probe_batches=(), jitter=None, priorities=None, schedule=None,
                 max_queued=2):
        """schedule: "round_robin" (call i -> replica i mod n) or "least_loaded" (the replica with the fewest unfinished frames,
        the higher-priority one on a tie, at most `max_queued` frames queued per replica -- the host waits for the oldest one
        beyond that).  Default: IA_PIPELINE_SCHEDULE, else least loaded when the streams differ in priority (the high-priority
        replica finishes its frames sooner and takes more of them: 600 -> 605 frames/s), round robin otherwise.  The frame ->
        replica mapping of the least-loaded schedule depends on timing; the frames do not (a replay takes its random draws from
        the process-wide generator in call order, whichever replica it runs on)."""
        import os
        self.replicas = [model] + [clone_for_stream(model) for _ in range(n_in_flight - 1)]
        env = [int(v) for v in os.environ.get("IA_STREAM_PRIORITIES", "").split(",") if v.strip()]
        if priorities is None:
            priorities = env or ([-1] + [0] * (n_in_flight - 1) if n_in_flight >= 3 else [0] * n_in_flight)
        self.priorities = [int(priorities[k % len(priorities)]) for k in range(n_in_flight)]
        self.streams = [torch.cuda.Stream(device=batch["rays_o"].device, priority=p) for p in self.priorities]
        self.graphs = []
        for m, s in zip(self.replicas, self.streams):
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self.graphs.append(GraphedRenderer(m, batch, img_size, margin=margin, probe_batches=probe_batches, jitter=jitter))
            torch.cuda.current_stream().wait_stream(s)
        self.events = [torch.cuda.Event() for _ in self.replicas]
        self.calls = 0
        self.schedule = schedule or os.environ.get("IA_PIPELINE_SCHEDULE", "") or ("least_loaded" if len(set(self.priorities)) > 1 else "round_robin")
        assert self.schedule in ("round_robin", "least_loaded"), self.schedule
        self.max_queued = int(os.environ.get("IA_PIPELINE_MAX_QUEUED", max_queued))
        self._unfinished = [[] for _ in self.replicas]     # per replica: events of the frames enqueued and not yet seen finished
        self._call_ids = [[] for _ in self.replicas]       # per replica: the global call number of its local calls
        self.frames_per_replica = [0 for _ in self.replicas]

    def _pick(self):
        n = len(self.graphs)
        if self.schedule == "round_robin":
            return self.calls % n
        for q in self._unfinished:
            while q and q[0].query():
                q.pop(0)
        k = min(range(n), key=lambda j: (len(self._unfinished[j]), self.priorities[j], (j - self.calls) % n))
        if len(self._unfinished[k]) >= self.max_queued:     # every replica has its queue full: wait for the oldest frame of this one
            self._unfinished[k].pop(0).synchronize()
        return k

    def __call__(self, batch, consume=None):
        """Launch one frame on the next replica's stream and return (outputs, replica index) without making any other
        stream wait.  `consume(outputs, k)`, if given, is executed right behind the frame ON THAT STREAM (reductions,
        copies into a frame buffer, encoding ...): the outputs are only ordered with respect to that stream; anybody else
        waits on `self.events[k]` (recorded behind `consume`) or calls `synchronize()`.  The replica's stream first waits
        for what the CALLER's stream has enqueued so far (the producers of `batch`; keep frame consumers off that stream,
        or the frames serialise behind them)."""
        k = self._pick()
        self._call_ids[k].append(self.calls)
        self.frames_per_replica[k] += 1
        self.streams[k].wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.streams[k]):
            out = self.graphs[k](batch)
            for key in ("global_orient", "body_pose", "transl"):   # read on this stream: tell the allocator, or a
                if torch.is_tensor(batch.get(key)):                                # freed input could be reused before the copy ran
                    batch[key].record_stream(self.streams[k])
            if consume is not None:
                consume(out, k)
            if self.schedule == "least_loaded":
                self.events[k] = torch.cuda.Event()
                self._unfinished[k].append(self.events[k])
            self.events[k].record()
        self.calls += 1
        return out, k

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def refresh_weights(self):
        """after the (shared) master weights changed: update every replica's fp16 shadow and MFMA fragments in place, on the
        replica's stream, so that the captured graphs render with the new weights from their next replay on"""
        for m, s in zip(self.replicas, self.streams):
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                m.net_coarse.refresh()

    @property
    def incomplete_calls(self):
        return sorted(self._call_ids[k][c] for k, g in enumerate(self.graphs) for c in g.incomplete_calls)

    def finish(self):
        return sum(g.finish() for g in self.graphs)


def build_synthetic_model(device, resolution=128, n_levels=16, max_samples=256, max_batch=291600, seed=42,
                          cano_pose="A_pose", blendshapes=False, betas=None, version=None):
    """Synthetic body + field wired into the three plugins.
    Returns (model, body dict, field dict).  blendshapes / betas: a body with non-zero shape and pose directions and a dense
    joint regressor (`synthetic.make_body(blendshapes=True)`) initialised with these shape coefficients -- what a real SMPL
    pickle + a dataset's betas are; the default is the zero-blendshape body of SURVEY.md 8d."""
    body = synthetic.make_body(seed, blendshapes=blendshapes)
    smpl = SMPL.from_dict(body).to(device)
    opt = dict(softmax_mode="hierarchical", resolution=resolution, cano_pose=cano_pose, precision=32)
    if version is not None:
        opt["version"] = version
    deformer = SNARFDeformer(None, "neutral", opt, body_model=smpl)
    betas = torch.zeros(1, 10, device=device) if betas is None else torch.as_tensor(
        np.asarray(betas, np.float32).reshape(1, 10), device=device)
    deformer.initialize(betas, device)
    deformer.initialized = True
    # canonical joints for the synthetic density (posed with the canonical pose)
    cano = smpl(betas=betas, body_pose=torch.as_tensor(synthetic.cano_pose(cano_pose), device=device)[None],
                return_verts=False).joints[0].cpu().numpy()
    fp = synthetic.make_field(cano, deformer.bbox.cpu().numpy(), seed=seed, n_levels=n_levels)
    net = NeRFNGPNet(dict(center=[0, -0.3, 0], scale=[2.5, 2.5, 2.5]), n_levels=n_levels).to(device)
    net.load_field_dict(fp)
    net.initialize(deformer.bbox)
    renderer = Raymarcher(max_samples, max_batch).to(device)
    renderer.initialize(1)
    model = AvatarModel(deformer, net, renderer).to(device)
    return model, body, fp


def make_batch(device, res, pose72, transl, betas=None):
    """One animate.py batch (animate.py:60-80): camera rays + SMPL parameters."""
    o, d = synthetic.make_camera_rays(res)
    pose72 = np.asarray(pose72, np.float32)
    transl = np.asarray(transl, np.float32)
    dist = float(np.sqrt((transl ** 2).sum()))
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=device)
    return {
        "rays_o": t(o)[None], "rays_d": t(d)[None],
        "betas": t(np.zeros((1, 10), np.float32) if betas is None else betas).reshape(1, 10),
        "global_orient": t(pose72[:3])[None], "body_pose": t(pose72[3:])[None], "transl": t(transl)[None],
        "near": torch.full((1, res * res), dist - 1, device=device),
        "far": torch.full((1, res * res), dist + 1, device=device),
    }
