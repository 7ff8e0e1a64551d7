"""Produce radet_amd/tune_gfx950.json: the autotuner's tile / split choices for the standard geometries, timed with
more repetitions than the start-up tuner (run on an otherwise idle MI355X).

    RADET_TUNE_REPS=9 python tools/make_tune.py gpurun_out/tune_gfx950.json
"""
import os
import sys

os.environ.setdefault("RADET_TUNE_REPS", "9")
os.environ["RADET_TUNE_FILE"] = "/nonexistent/none.json"        # start from nothing but what this run measures
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from radet_amd import kernels as K
from radet_amd.models import build_detector
from radet_amd.utils import Config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--retune-strided" in sys.argv:
    # keep the shipped choices except those of the 3x3 / 2 convs (their dgrad changed to one class launch): timed again
    K.load_tune_cache()
    for key in [k for k in K._TUNE_CACHE if k[0][3] == 3 and k[0][4] > 1]:
        del K._TUNE_CACHE[key]
elif "--retune-x3" in sys.argv:
    # keep the shipped choices except the entries of the default fp32 arithmetic (new tile candidates): timed again
    K.load_tune_cache()
    for cache in (K._TUNE_CACHE, K._WTUNE_CACHE):
        for key in [k for k in cache if k[-1] == "x3"]:
            del cache[key]
elif "--retune-h2" in sys.argv:
    # keep the shipped choices except the entries of the fp16 hi / lo arithmetic (their prologues changed): timed again
    K.load_tune_cache()
    for cache in (K._TUNE_CACHE, K._WTUNE_CACHE):
        for key in [k for k in cache if "h2" in k]:
            del cache[key]
elif os.path.exists(K._PACKAGED_TUNE) and "--keep" not in sys.argv:
    K._TUNE_LOADED = True                                       # ignore the shipped file too
out = [a for a in sys.argv[1:] if not a.startswith("--")][0]
JOBS = [  # depth, math, (B, H, W) list
    (50, "fp32", [(4, 480, 640), (1, 480, 640), (2, 480, 640), (8, 480, 640), (16, 480, 640)]),   # products from bf16 planes
    (50, "fp32-mfma", [(4, 480, 640), (1, 480, 640), (2, 480, 640), (8, 480, 640)]),     # native fp32 MFMA
    (50, "bf16-storage", [(4, 480, 640), (8, 480, 640), (1, 480, 640)]),
    (50, "bf16", [(4, 480, 640), (8, 480, 640)]),
    (101, "fp32", [(2, 800, 800)]),
    (101, "fp32-mfma", [(2, 800, 800)]),
    (101, "bf16-storage", [(2, 800, 800)]),
]
if "--retune-x3" in sys.argv or "--retune-h2" in sys.argv or "--fp32-only" in sys.argv:
    JOBS = [j for j in JOBS if j[1] == "fp32"]
for depth, math, geos in JOBS:
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "bop", "r50_ycbv_pbr.py"))
    cfg.model["pretrained"] = None
    cfg.model["backbone"]["depth"] = depth
    det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
    rt = det.runtime(math=math)
    for (B, H, W) in geos:
        rt.engine.prepare(B, H, W)
        torch.cuda.synchronize()
        print(f"r{depth} {math} B={B} {W}x{H}: {len(K._TUNE_CACHE)} igemm / {len(K._WTUNE_CACHE)} wgrad entries", flush=True)
    del rt, det
    torch.cuda.empty_cache()
K._TUNE_DIRTY = True
K.save_tune_cache(out)
print("wrote", out, os.path.getsize(out), "bytes")
