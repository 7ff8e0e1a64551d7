"""Registry / build_from_cfg with the semantics the reference gets from mmcv.utils
(radet/models/builder.py:1-35; mmcv is not a dependency here)."""
import inspect


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return key in self._module_dict

    def __repr__(self):
        return f"Registry(name={self._name}, items={sorted(self._module_dict)})"

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def _register(self, cls, name=None, force=False):
        if not inspect.isclass(cls):
            raise TypeError(f"module must be a class, but got {type(cls)}")
        name = name or cls.__name__
        if not force and name in self._module_dict:
            raise KeyError(f"{name} is already registered in {self._name}")
        self._module_dict[name] = cls

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register(module, name, force)
            return module

        def _deco(cls):
            self._register(cls, name, force)
            return cls
        return _deco


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f"cfg must be a dict, but got {type(cfg)}")
    if "type" not in cfg and not (default_args and "type" in default_args):
        raise KeyError(f'`cfg` or `default_args` must contain the key "type", but got {cfg}')
    if not isinstance(registry, Registry):
        raise TypeError(f"registry must be a Registry, but got {type(registry)}")
    args = dict(cfg)
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop("type")
    if isinstance(obj_type, str):
        cls = registry.get(obj_type)
        if cls is None:
            raise KeyError(f"{obj_type} is not in the {registry.name} registry")
    elif inspect.isclass(obj_type):
        cls = obj_type
    else:
        raise TypeError(f"type must be a str or valid type, but got {type(obj_type)}")
    return cls(**args)
