import importlib


def instantiate(cfg, *args, **kwargs):
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    mod, name = target.rsplit(".", 1)
    fn = getattr(importlib.import_module(mod), name)
    cfg.update(kwargs)
    return fn(*args, **cfg)
