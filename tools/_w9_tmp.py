import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radet_amd import kernels as K
from radet_amd.kernels import ConvGeom, Levels, Planes
from tools.bench_p3 import timeit
dev = torch.device("cuda")
lv = Levels([(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)], 4)
g = ConvGeom(lv, 256, 256, 3, 1, 1); g.x3 = "h2"
x = torch.relu(torch.randn(lv.rows, 256, device=dev)); dy = torch.randn(lv.rows, 256, device=dev) * 1e-3
xp, dyp = Planes.from_float(x, kind="h2"), Planes.from_float(dy, kind="h2")
S = 16
g.nsplit = S
slabs = torch.empty(S, 256, 9, 256, device=dev); bp = torch.empty(S, 256, device=dev)
for win in ((False, True) if os.environ.get("BOTH") else (False,)):
    K.WGRAD9_WINDOWS = win
    us = min(timeit(lambda: K.conv_wgrad(g, dyp, xp, slabs, bp), n=30) for _ in range(3))
    print(f"dbg={os.environ.get('RADET_DBG_WGRAD', '0')} windows={win}: {us:7.1f} us", flush=True)
