"""Option access that works for dicts, omegaconf DictConfig and plain objects
(the reference passes hydra DictConfigs: confs/deformer/fast_snarf.yaml:4-8)."""


def get(opt, key, default=None):
    if opt is None:
        return default
    if isinstance(opt, dict):
        return opt.get(key, default)
    if hasattr(opt, "get"):
        try:
            return opt.get(key, default)
        except TypeError:
            pass
    return getattr(opt, key, default)
