"""Import-path alias: `radet.models`, `radet.core`, `radet.ops`, `radet.datasets`, `radet.apis`, `radet.utils` (and their
sub-modules, e.g. `radet.datasets.pipelines`, `radet.models.losses`) resolve to the MI355X implementation in `radet_amd`,
so that code written against the reference's package name runs unchanged (`from radet.models import build_detector`,
`from radet.ops import vote_nms`, `from radet.core import bbox_overlaps`, ...)."""
import importlib
import sys

import radet_amd
from radet_amd import __version__  # noqa: F401

for _name in ("models", "core", "ops", "datasets", "utils", "apis"):
    importlib.import_module(f"radet_amd.{_name}")
for _full, _mod in list(sys.modules.items()):            # every loaded radet_amd.x.y is also radet.x.y (the same module object)
    if _full.startswith("radet_amd.") and _mod is not None:
        sys.modules[__name__ + _full[len("radet_amd"):]] = _mod
for _name in ("models", "core", "ops", "datasets", "utils", "apis"):
    setattr(sys.modules[__name__], _name, sys.modules[f"radet_amd.{_name}"])
