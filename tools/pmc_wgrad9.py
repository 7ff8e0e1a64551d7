#!/usr/bin/env python
"""The towers' all-taps weight gradient on plane operands (conv_wgrad9p_kernel), N launches alone on the device -- the target of
`PMC_PROG=tools/pmc_wgrad9.py bash tools/pmc_one.sh <out>` (counter passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radet_amd import kernels as K  # noqa: E402
from radet_amd.kernels import ConvGeom, Levels, Planes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lv = Levels([(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)], 4)
g = ConvGeom(lv, 256, 256, 3, 1, 1)
g.x3 = True
torch.manual_seed(1)
x = Planes.from_float(torch.relu(torch.randn(lv.rows, 256, device="cuda")))
dy = Planes.from_float(torch.randn(g.lout.rows, 256, device="cuda") * 0.01)
slabs = torch.empty(g.nsplit * 256 * 9 * 256, device="cuda")
for _ in range(n):
    K.conv_wgrad(g, dy, x, slabs)
torch.cuda.synchronize()
