from .config import Config, ConfigDict, to_config_dict
from .registry import Registry, build_from_cfg

__all__ = ["Config", "ConfigDict", "to_config_dict", "Registry", "build_from_cfg"]
