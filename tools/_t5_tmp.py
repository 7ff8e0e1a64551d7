import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radet_amd import kernels as K
from radet_amd.kernels import ConvGeom, Levels, Planes
from tools.bench_p3 import timeit
dev = torch.device("cuda")
lv = Levels([(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)], 4)
g = ConvGeom(lv, 256, 256, 3, 1, 1); g.x3 = "h2"
x = torch.relu(torch.randn(lv.rows, 256, device=dev)); w = torch.randn(256 * 9, 256, device=dev) * 0.05
xp, wp = Planes.from_float(x, kind="h2"), Planes.from_float(w, kind="h2")
y = torch.empty(lv.rows, 256, device=dev)
for t in [6 | (1 << 12), 6 | (1 << 12) | K.STAGES3, 5 | (1 << 12), 5 | (1 << 12) | K.STAGES3, 1 | (1 << 12), 1 | (1 << 12) | K.STAGES3]:
    us = min(timeit(lambda: K.conv_fwd(g, xp, wp, None, y, tile=t), n=30) for _ in range(3))
    print(f"tile {t:#x}: {us:8.1f} us", flush=True)
