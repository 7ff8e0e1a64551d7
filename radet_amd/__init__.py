"""radet_amd -- MI355X-native implementation of RADet's detector hot path.

Importing the package never touches the GPU; the HIP library (radet_amd/libradet_hip.so) is loaded
on first use and its absence is a hard error (no CPU fallback)."""
__version__ = "0.1.0"
