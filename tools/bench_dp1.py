#!/usr/bin/env python
"""The data-parallel step on ONE GPU: a 1-rank RCCL process group with the bucketed exchange forced (RADET_FORCE_REDUCER=1), so
that the collectives' stream(s) are in play next to the engine's four -- step time against the plain single-GPU step.
    python tools/bench_dp1.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import bench
from radet_amd.models import build_detector
from radet_amd.utils import Config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29547")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
cfg = Config.fromfile(os.path.join(ROOT, "configs", "bop", "r50_ycbv_pbr.py"))
cfg.model["pretrained"] = None
torch.manual_seed(0)
det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
rt = det.runtime()
rt.init_optimizer()
img, boxes, labels, p2g, pw = bench.make_batch(0, 4, torch.device("cuda", 0))
tg = rt.pack_targets([torch.from_numpy(b) for b in boxes], [torch.from_numpy(l) for l in labels], list(p2g), list(pw))
for force in ("0", "1", "0", "1"):
    os.environ["RADET_FORCE_REDUCER"] = force
    for _ in range(5):
        rt.train_step(img, tg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        rt.train_step(img, tg)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"bucketed RCCL exchange {'on (1 rank)' if force == '1' else 'off'}: {dt * 1e3:.3f} ms/step  {4 / dt:.1f} images/s", flush=True)
dist.destroy_process_group()
