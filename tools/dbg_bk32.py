import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radet_amd import kernels as K
def run(hw, B, cin, cout, k, tile):
    lv = K.Levels(hw, B)
    g = K.ConvGeom(lv, cin, cout, k, 1, k // 2)
    x = torch.randn(lv.rows, cin, device="cuda")
    w = torch.randn(cout, k * k, cin, device="cuda") * 0.05
    y1 = torch.empty(lv.rows, cout, device="cuda"); y2 = torch.empty_like(y1)
    K.conv_fwd(g, x, w, None, y1, tile=tile & 0xFF, splitk=False)
    K.conv_fwd(g, x, w, None, y2, tile=tile, splitk=False)
    d = (y1 - y2).abs()
    bad_rows = (d.max(1).values > 1e-3).nonzero().reshape(-1)
    print(hw, B, cin, cout, k, hex(tile), "maxdiff", d.max().item(), "bad rows", bad_rows.numel(), bad_rows[:8].tolist(), bad_rows[-4:].tolist())
cases = [([(8, 16)], 1, 256, 64, 1), ([(8, 16)], 1, 64, 64, 3), ([(8, 16)], 1, 256, 256, 3), ([(60, 80)], 1, 256, 256, 3),
         ([(60, 80), (30, 40)], 2, 256, 256, 3)]
c = cases[int(sys.argv[1])]
run(*c, int(sys.argv[2], 0))
