#!/usr/bin/env python
"""The head-tower forward GEMM on fp16 plane pairs (256 x 128 tile, as in the step), N launches alone on the device -- the target
of `PMC_PROG=tools/pmc_tower.py bash tools/pmc_one.sh <out>` (stall / LDS counters of the kernel that is 36 % of the conv time).
    python tools/pmc_tower.py [tile, default 6] [n]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radet_amd import kernels as K  # noqa: E402

tile = int(sys.argv[1], 0) if len(sys.argv) > 1 else 6        # e.g. 6, or 0x20006 for three LDS stages
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lv = K.Levels([(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)], 4)
g = K.ConvGeom(lv, 256, 256, 3, 1, 1)
g.x3 = "h2"
torch.manual_seed(0)
x = torch.relu(torch.randn(lv.rows, 256, device="cuda"))
w = torch.randn(256 * 9, 256, device="cuda") * 0.02
xq, wq = K.Planes.from_float(x, kind="h2"), K.Planes.from_float(w, kind="h2")
y = torch.empty(lv.rows, 256, device="cuda")
for _ in range(n):
    K.conv_fwd(g, xq, wq, None, y, tile=tile | (1 << 12))
torch.cuda.synchronize()
print("ok", float(y.abs().max()))
