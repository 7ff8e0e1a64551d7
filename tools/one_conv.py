"""Run one conv shape/tile a few times (for rocprofv3 --pmc / debugging). Usage: one_conv.py tile [M-kind]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radet_amd import kernels as K
tile = int(sys.argv[1], 0) if len(sys.argv) > 1 else 1
kind = sys.argv[2] if len(sys.argv) > 2 else "big"
hw = [(128, 256)] if kind == "big" else [(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)]
lv = K.Levels(hw, 4)
g = K.ConvGeom(lv, 256, 256, 3, 1, 1)
x = torch.randn(lv.rows, 256, device="cuda")
w = torch.randn(256, 9, 256, device="cuda") * 0.05
y = torch.empty(lv.rows, 256, device="cuda")
y2 = torch.empty(lv.rows, 256, device="cuda")
dy = torch.randn(lv.rows, 256, device="cuda")
slabs = torch.empty(g.nsplit * 256 * 9 * 256, device="cuda")
K.conv_fwd(g, x, w, None, y2, relu=True, tile=1)
for _ in range(5):
    K.conv_fwd(g, x, w, None, y, relu=True, tile=tile)
    if len(sys.argv) > 3:
        K.conv_wgrad(g, dy, x, slabs, None)
torch.cuda.synchronize()
print(hex(tile), "max diff vs tile 1:", (y - y2).abs().max().item())
