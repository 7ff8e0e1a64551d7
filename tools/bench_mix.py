#!/usr/bin/env python
"""Stream-budget check: train steps, then streamed inference (and a captured detect graph), then train steps again in ONE
process -- the second training block must run at the speed of the first (a fifth HIP stream created anywhere in between would
cost it ~35 %, radet_amd/engine.py "stream budget").
    python tools/bench_mix.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from radet_amd.models import build_detector
from radet_amd.utils import Config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = Config.fromfile(os.path.join(ROOT, "configs", "bop", "r50_ycbv_pbr.py"))
cfg.model["pretrained"] = None
torch.manual_seed(0)
det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
rt = det.runtime()
rt.init_optimizer()
img, boxes, labels, p2g, pw = bench.make_batch(0, 4, torch.device("cuda", 0))
tg = rt.pack_targets([torch.from_numpy(b) for b in boxes], [torch.from_numpy(l) for l in labels], list(p2g), list(pw))


def train_block(tag, n=20):
    det.train()
    for _ in range(5):
        rt.train_step(img, tg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        rt.train_step(img, tg)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{tag}: {dt * 1e3:.3f} ms/step", flush=True)
    return dt


a = train_block("train, fresh process")
det.eval()
B = 8
imgs = torch.randn(B, 3, 480, 640).cuda()
metas = [dict(img_shape=(480, 640, 3), scale_factor=np.ones(4, np.float32)) for _ in range(B)]
n = sum(len(o) for o in rt.detect_stream(((imgs, metas) for _ in range(6)), det.test_cfg, rescale=True))
rt.detect_graph(imgs[:1], metas[:1], det.test_cfg, rescale=True)
rt.detect_graph(imgs[:1], metas[:1], det.test_cfg, rescale=True)
print(f"streamed inference over {n} images + a captured single-image graph", flush=True)
b = train_block("train, after inference")
assert b < 1.1 * a, (a, b)
print("ok")
