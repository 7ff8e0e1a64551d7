import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radet_amd import kernels as K
from radet_amd.kernels import ConvGeom, Levels, Planes
from tools.bench_p3 import timeit
dev = torch.device("cuda")
lv = Levels([(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)], 4)
g = ConvGeom(lv, 256, 256, 3, 1, 1); g.x3 = True
x = torch.relu(torch.randn(lv.rows, 256, device=dev)); w = torch.randn(256 * 9, 256, device=dev) * 0.05
xp, wp = Planes.from_float(x), Planes.from_float(w)
y1, y2 = torch.empty(lv.rows, 256, device=dev), torch.empty(lv.rows, 256, device=dev)
flop = 2.0 * lv.rows * 256 * 256 * 9 * 2
tiles = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [6, 5]
for t in tiles:
    us = timeit(lambda: K.conv_fwd_pair(g, dict(x=xp, w=wp, y=y1), dict(x=xp, w=wp, y=y2), tile=t))
    print(f"dbg={os.environ.get('RADET_DBG_IGEMM','0'):>2s} pair tile {t}: {us:8.1f} us {flop/us/1e6:7.1f} TFLOP/s", flush=True)
