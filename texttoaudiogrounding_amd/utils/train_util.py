"""Object construction by dotted path -- the plugin mechanism of the reference
(utils/train_util.py:120-137): ``{"type": "pkg.mod.Class", "args": {...}}`` with nested dicts
instantiated recursively."""
import importlib


def get_obj_from_str(string):
    module, cls = string.rsplit(".", 1)
    return getattr(importlib.import_module(module), cls)


def init_obj_from_str(config, **kwargs):
    args = dict(config.get("args", {}))
    args.update(kwargs)
    for k, v in config.items():
        if k not in ("type", "args") and isinstance(v, dict) and k not in kwargs:
            args[k] = init_obj_from_str(v)
    return get_obj_from_str(config["type"])(**args)
