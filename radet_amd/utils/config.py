"""Python-file configs with `_base_` inheritance and attribute access (the subset of mmcv.Config the
hot path uses: tools/train.py:90-92, atss_head.py:45, vote_wrapper.py:8 in the reference)."""
import copy
import os


class ConfigDict(dict):
    """dict with attribute access; missing attributes raise AttributeError, .get works as usual."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = value

    def copy(self):
        return ConfigDict(self)

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_config_dict(obj):
    if isinstance(obj, dict):
        return ConfigDict({k: to_config_dict(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_config_dict(v) for v in obj)
    return obj


def _merge(base, child):
    out = dict(base)
    for k, v in child.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get("_delete_", False):
            out[k] = _merge(out[k], v)
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != "_delete_"}
            out[k] = v
    return out


def _load_file(path):
    ns = {}
    with open(path) as f:
        exec(compile(f.read(), path, "exec"), ns)
    cfg = {k: v for k, v in ns.items() if not k.startswith("__") and not callable(v) and not isinstance(v, type(os))}
    bases = cfg.pop("_base_", [])
    if isinstance(bases, str):
        bases = [bases]
    merged = {}
    for b in bases:
        merged = _merge(merged, _load_file(os.path.join(os.path.dirname(path), b)))
    return _merge(merged, cfg)


class Config:
    def __init__(self, cfg_dict=None, filename=None):
        super().__setattr__("_cfg_dict", to_config_dict(cfg_dict or {}))
        super().__setattr__("filename", filename)

    @staticmethod
    def fromfile(filename):
        return Config(_load_file(os.path.abspath(filename)), filename=filename)

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __contains__(self, name):
        return name in self._cfg_dict

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def merge_from_dict(self, options):
        nested = {}
        for full_key, v in options.items():
            d = nested
            keys = full_key.split(".")
            for k in keys[:-1]:
                d = d.setdefault(k, {})
            d[keys[-1]] = v
        super().__setattr__("_cfg_dict", to_config_dict(_merge(self._cfg_dict, nested)))

    @property
    def pretty_text(self):
        import pprint
        return pprint.pformat(dict(self._cfg_dict))
