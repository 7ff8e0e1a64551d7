import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # see radet_amd/__init__.py: main / side / RCCL streams, one HW queue each
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")

# Tile / split choices of the geometries only the TESTS use (odd sizes, small images, R101 at 200 x 264 ...) are pinned like the
# standard geometries' are in radet_amd/tune_gfx950.json: without this the start-up tuner times them on the spot in every run,
# its picks follow the box's timing noise, and with them the fp32 summation orders and which knife-edge ReLU masks flip --
# tests that compare against the fp32 oracle then see another set of flips in every run (round 6: one of them moved past its
# median bound with a new set of picks).  The file holds what one full `pytest -m gpu` run on an MI355X picked
# (`RADET_TUNE_FILE=... pytest -m gpu`, entries not in the packaged file); geometries it does not know are still tuned live.
# A copy in a scratch directory is used so that a run never edits the tracked file.
if "RADET_TUNE_FILE" not in os.environ:
    import shutil
    import tempfile
    _src = os.path.join(GOLDEN, "tune_tests_gfx950.json")
    if os.path.exists(_src):
        _dst = os.path.join(tempfile.gettempdir(), f"radet_tune_tests_{os.getpid()}.json")
        shutil.copyfile(_src, _dst)
        os.environ["RADET_TUNE_FILE"] = _dst


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load
