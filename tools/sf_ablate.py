"""Ablation of the split-at-fill tile on one large shape: RADET_DBG_IGEMM=<bits> python tools/sf_ablate.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radet_amd import kernels as K
from radet_amd.kernels import ConvGeom, Levels
from tools.bench_p3 import timeit
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for name, hw, cin, cout, k in (("fpn out", (60, 80), 256, 256, 3), ("layer3 down", (30, 40), 1024, 256, 1), ("layer2 up", (60, 80), 128, 512, 1)):
    lv = Levels([hw], B); g = ConvGeom(lv, cin, cout, k, 1, k // 2); g.x3 = True
    x = torch.relu(torch.randn(lv.rows, cin, device="cuda")); w = torch.randn(cout * k * k * cin, device="cuda") * 0.05
    y = torch.empty(g.lout.rows, cout, device="cuda")
    for t in (9 | (1 << 12), 1 | 0x200 | (1 << 12)):
        us = timeit(lambda: K.conv_fwd(g, x, w, None, y, relu=True, tile=t), n=10, warm=2)
        print(f"dbg={os.environ.get('RADET_DBG_IGEMM', '0')} {name} M={g.lout.rows} tile {t & 0xff}: {us:.1f} us {2.0 * g.lout.rows * cout * cin * k * k / us / 1e6:.1f} TF")
