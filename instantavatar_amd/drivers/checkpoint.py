"""Checkpoint I/O in the layout pytorch_lightning writes for DNeRFModel (train.py:15-21): a dict with
`state_dict` (keys `net_coarse.encoder.params`, `net_coarse.color_net.params`, `net_coarse.center`,
`net_coarse.scale`, `renderer.density_grid_{train,test}.*`, ... -- SURVEY.md section 5), `global_step`
and `epoch`.  Keys of modules that are not on the hot path (e.g. `SMPLParamEmbedding`, LPIPS) are
reported and ignored on load."""
import torch


def save_checkpoint(model, path, optimizer=None, epoch=0, scheduler=None):
    ckpt = {"state_dict": model.state_dict(), "global_step": int(getattr(model, "global_step", 0)), "epoch": int(epoch)}
    if optimizer is not None:
        ckpt["optimizer_states"] = [optimizer.state_dict()]
    if scheduler is not None:
        ckpt["lr_schedulers"] = [scheduler.state_dict()]
    torch.save(ckpt, path)
    return path


def load_checkpoint(model, path, map_location="cpu", strict_path_keys=True, optimizer=None, scheduler=None, skip_prefixes=(),
                    strict_self_check=False):
    """Returns (missing, unexpected).  The tcnn parameter vectors must have the tcnn-v1.6 layout this
    package restates ([MLP weights..., grid]); a size mismatch raises.  `optimizer` / `scheduler`: restored
    from Lightning's `optimizer_states[0]` / `lr_schedulers[0]` when the checkpoint holds them (resume).
    `skip_prefixes`: entries whose key starts with one of them stay at the model's own values (eval.py:64-67 keeps the freshly
    built `SMPL_param` tables of the test frames and takes everything else from the checkpoint).
    `strict_self_check`: a loaded encoder that fails `NeRFNGPNet.self_check` (non-finite features or a constant level: what a
    mis-laid-out table looks like) raises instead of warning.  The drivers that RENDER from a checkpoint (animate, novel_view,
    eval) pass True -- garbage frames with exit code 0 are worse than a refusal; tooling that inspects checkpoints keeps False."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    net = getattr(model, "net_coarse", None)
    enc = sd.get("net_coarse.encoder.params")
    if enc is not None and hasattr(net, "adopt_tcnn_layout") and enc.numel() != net.encoder.params.numel():
        # a checkpoint written with the OTHER of tcnn's two possible level tables: follow it (decided by the vector's size)
        r3 = net.adopt_tcnn_layout(enc.numel())     # resizes `encoder.params` in place: the Parameter object is kept
        print("load_checkpoint: encoder.params has %d elements -> tcnn layout with level-3 resolution %d adopted" % (enc.numel(), r3))
        if optimizer is not None:                   # moments of the old shape must not meet a gradient of the new one
            optimizer.state.pop(net.encoder.params, None)
    own = model.state_dict()
    take = {}
    unexpected = []
    for k, v in sd.items():
        if any(k.startswith(p) for p in skip_prefixes):
            continue
        if k in own:
            if tuple(own[k].shape) != tuple(v.shape):
                hint = ""
                net = getattr(model, "net_coarse", None)
                if k.endswith("encoder.params") and hasattr(net, "tcnn_encoder_sizes"):
                    sizes = net.tcnn_encoder_sizes(net.n_levels, net.log2_T)
                    other = [r for r, n in sizes.items() if n == v.numel()]
                    if other:
                        hint = (" -- the checkpoint uses the tcnn layout with level-3 resolution %d; build the network with "
                                "level3_res=%d (or IA_TCNN_LEVEL3_RES=%d)" % (other[0], other[0], other[0]))
                raise ValueError("checkpoint tensor %s has shape %s, expected %s%s" % (k, tuple(v.shape), tuple(own[k].shape), hint))
            take[k] = v
        else:
            unexpected.append(k)
    missing = [k for k in own if k not in take and not any(k.startswith(p) for p in skip_prefixes)]
    on_path = [k for k in missing if k.startswith("net_coarse.") and k.endswith("params")]
    if strict_path_keys and on_path:
        raise KeyError("checkpoint lacks the field parameters: %s" % on_path)
    model.load_state_dict(take, strict=False)
    if net is not None:
        if hasattr(net, "mark_updated"):
            net.mark_updated()
        if hasattr(net, "self_check") and net.encoder.params.is_cuda and "net_coarse.encoder.params" in take:
            # finite, no constant level: a shifted layout would show here.  A REPORT, not a gate: the state is already
            # committed at this point, and a legitimately degenerate field (tiny coarse levels, a collapsed bbox) must stay
            # loadable -- callers that want a hard failure call net.self_check() themselves
            try:
                model.tcnn_self_check = net.self_check()
            except ValueError as e:
                import warnings
                model.tcnn_self_check = {"failed": str(e)}
                if strict_self_check:
                    raise
                warnings.warn("load_checkpoint: %s" % e)
    for g in ("density_grid_train", "density_grid_test"):
        grid = getattr(getattr(model, "renderer", None), g, None)
        if grid is not None and hasattr(grid, "pack_bits") and grid.density_field.is_cuda:
            grid.pack_bits()  # the kernels read the bit-packed mirror of density_field
    model.global_step = int(ckpt.get("global_step", 0)) if isinstance(ckpt, dict) else 0
    if optimizer is not None and isinstance(ckpt, dict) and ckpt.get("optimizer_states"):
        optimizer.load_state_dict(ckpt["optimizer_states"][0])
    if scheduler is not None and isinstance(ckpt, dict) and ckpt.get("lr_schedulers"):
        scheduler.load_state_dict(ckpt["lr_schedulers"][0])
    return missing, unexpected
