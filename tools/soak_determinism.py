#!/usr/bin/env python
"""Soak test of the multi-stream step: two detectors from the same seed run N train steps each (with streamed inference in
between every 100 steps); losses of every step and all parameters / AdamW moments at the end must be bit-identical -- a race
between the tower chains, the weight-gradient streams or the borrowed chain stream would show up as a difference.
    python tools/soak_determinism.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from radet_amd.models import build_detector
from radet_amd.utils import Config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
img, boxes, labels, p2g, pw = bench.make_batch(0, 4, dev)
metas = [dict(img_shape=(480, 640, 3), scale_factor=np.ones(4, np.float32)) for _ in range(4)]


def run():
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "bop", "r50_ycbv_pbr.py"))
    cfg.model["pretrained"] = None
    torch.manual_seed(0)
    det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
    rt = det.runtime()
    rt.init_optimizer()
    tg = rt.pack_targets([torch.from_numpy(b) for b in boxes], [torch.from_numpy(l) for l in labels], list(p2g), list(pw))
    hist, ndet = [], 0
    for it in range(steps):
        hist.append(rt.train_step(img, tg).clone())
        if it % 100 == 99:
            det.eval()
            ndet += sum(int(d.shape[0]) for out in rt.detect_stream(((img, metas) for _ in range(3)), det.test_cfg) for d, _ in out)
            det.train()
    torch.cuda.synchronize()
    return torch.stack(hist).cpu(), rt.flat.params.clone().cpu(), rt.opt_state["m"].clone().cpu(), rt.opt_state["v"].clone().cpu(), ndet


a, b = run(), run()
assert torch.isfinite(a[0]).all()
for name, x, y in zip(("losses of every step", "parameters", "AdamW m", "AdamW v"), a[:4], b[:4]):
    same = torch.equal(x, y)
    print(f"{name}: {'bit-identical' if same else 'DIFFER'} ({x.numel()} values)")
    assert same, name
assert a[4] == b[4]
print(f"ok: {steps} steps twice, {a[4]} detections in the interleaved inference passes; last losses {a[0][-1].tolist()}")
