"""Minimal stand-in for the part of Hydra the reference uses (train.py:10-28, animate.py:84-92):
read `confs/<group>/<name>.yaml`, resolve `${a.b}` interpolations against a context, build the
object named by `_target_` (hydra.utils.instantiate with `_recursive_=False`: nested dicts stay
dicts, exactly what DNeRFModel.__init__ passes on, DNeRF.py:22-28)."""
import importlib
import os
import re

import yaml

_INTERP = re.compile(r"\$\{([^}]+)\}")


def _lookup(ctx, dotted):
    cur = ctx
    for part in dotted.split("."):
        cur = cur[part]
    return cur


def resolve(node, ctx):
    """Resolve `${x.y}` recursively; a string that IS one interpolation keeps the value's type."""
    if isinstance(node, dict):
        return {k: resolve(v, ctx) for k, v in node.items()}
    if isinstance(node, list):
        return [resolve(v, ctx) for v in node]
    if isinstance(node, str):
        m = _INTERP.fullmatch(node.strip())
        if m:
            return _lookup(ctx, m.group(1))
        return _INTERP.sub(lambda mm: str(_lookup(ctx, mm.group(1))), node)
    return node


def load_group(conf_dir, group, name, ctx):
    with open(os.path.join(conf_dir, group, name + ".yaml")) as f:
        return resolve(yaml.safe_load(f), ctx)


def instantiate(node, **extra):
    """hydra.utils.instantiate(node, _recursive_=False)"""
    node = dict(node)
    target = node.pop("_target_")
    mod, _, attr = target.rpartition(".")
    cls = getattr(importlib.import_module(mod), attr)
    node.update(extra)
    return cls(**node)


def build_plugins(conf_dir, deformer="fast_snarf", network="ngp", renderer="raymarcher_acc", gender="neutral",
                  precision=32, deformer_kwargs=None):
    """The three plugin objects of DNeRFModel.__init__ (DNeRF.py:22-28) from the conf groups."""
    ctx = {"dataset": {"gender": gender}, "train": {"precision": precision}}
    d = load_group(conf_dir, "deformer", deformer, ctx)
    n = load_group(conf_dir, "network", network, ctx)
    r = load_group(conf_dir, "renderer", renderer, ctx)
    return instantiate(d, **(deformer_kwargs or {})), instantiate(n), instantiate(r)
