"""Import-path alias: `radet.models`, `radet.core`, `radet.ops`, `radet.datasets` resolve to the MI355X
implementation in `radet_amd` so that code written against the reference's package name runs unchanged
(`from radet.models import build_detector`, `from radet.ops import vote_nms`, ...)."""
import importlib
import sys

import radet_amd
from radet_amd import __version__  # noqa: F401

for _name in ("models", "core", "ops", "datasets", "utils"):
    _mod = importlib.import_module(f"radet_amd.{_name}")
    sys.modules[f"{__name__}.{_name}"] = _mod
    setattr(sys.modules[__name__], _name, _mod)
