"""radet_amd -- MI355X-native implementation of RADet's detector hot path.

Importing the package never touches the GPU; the HIP library (radet_amd/libradet_hip.so) is loaded
on first use and its absence is a hard error (no CPU fallback)."""
__version__ = "0.1.0"

import os as _os

# The engine keeps up to three HIP streams busy (main, side: wgrad / slab reduction, the process group's RCCL
# stream).  HIP maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues; when two busy streams share a
# queue their barrier packets serialise them.  Ask for 8 queues unless the user chose otherwise -- read by the HIP
# runtime when it initialises, i.e. effective when radet_amd is imported before the first GPU call.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
