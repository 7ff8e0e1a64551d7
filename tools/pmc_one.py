#!/usr/bin/env python
"""One conv geometry, one tile, N launches -- the target of a `rocprofv3 --pmc ...` pass (see tools/pmc_one.sh).
    python tools/pmc_one.py <H> <W> <cin> <cout> <k> <stride> <tile_override hex> [n]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radet_amd import kernels as K  # noqa: E402

H, W, cin, cout, k, stride = (int(v) for v in sys.argv[1:7])
tile = int(sys.argv[7], 16)
n = int(sys.argv[8]) if len(sys.argv) > 8 else 20
lv = K.Levels([(H, W)], 4)
g = K.ConvGeom(lv, cin, cout, k, stride, k // 2)
g.x3 = True
torch.manual_seed(0)
x = torch.relu(torch.randn(lv.rows, cin, device="cuda"))
w = torch.randn(cout * k * k * cin, device="cuda") * 0.05
y = torch.empty(g.lout.rows, cout, device="cuda")
for _ in range(n):
    K.conv_fwd(g, x, w, None, y, relu=True, tile=tile)
torch.cuda.synchronize()
