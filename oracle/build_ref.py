"""Build the REAL reference C++ ops into oracle/_ref/ (test infrastructure only).

The three reference extensions are single-file C++ torch extensions
(/root/reference/radet/ops/{vote/vote_ext.cpp, cluster/cluster_ext.cpp,
bbox2distance/bbox2distance_ext.cpp}); they
compile as-is with g++ against the torch headers of this image.  Sources are
compiled where they lie (never copied); only the resulting .so files land in
oracle/_ref/, which is git-ignored but travels to the GPU box with gpurun.

Nothing under radet_amd/ may import this module: it exists so that tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg can check / time the
oracle and the HIP path against the reference's own native code.
"""
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")
REFERENCE_ROOT = "/root/reference"

_SOURCES = {
    "ref_vote_ext": "radet/ops/vote/vote_ext.cpp",
    "ref_cluster_ext": "radet/ops/cluster/cluster_ext.cpp",
    "ref_bbox2distance_ext": "radet/ops/bbox2distance/bbox2distance_ext.cpp",
}


def build(verbose=False):
    """Compile the reference ops if /root/reference is present. Returns list of built names."""
    if not os.path.isdir(REFERENCE_ROOT):
        return []
    from torch.utils import cpp_extension
    os.makedirs(REF_DIR, exist_ok=True)
    built = []
    for name, rel in _SOURCES.items():
        so = os.path.join(REF_DIR, name + ".so")
        src = os.path.join(REFERENCE_ROOT, rel)
        if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(src):
            built.append(name)
            continue
        cpp_extension.load(name=name, sources=[src], build_directory=REF_DIR,
                           extra_cflags=["-O2"], verbose=verbose, is_python_module=True)
        built.append(name)
    return built


def load(name):
    """Import a prebuilt reference module from oracle/_ref (None if absent)."""
    so = os.path.join(REF_DIR, name + ".so")
    if not os.path.exists(so):
        return None
    if name in sys.modules:
        return sys.modules[name]
    import torch  # noqa: F401  (libtorch must be loaded first)
    spec = importlib.util.spec_from_file_location(name, so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules[name] = mod
    return mod


if __name__ == "__main__":
    print(build(verbose=True))
