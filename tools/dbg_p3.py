import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radet_amd import kernels as K
from radet_amd.kernels import ConvGeom, Levels, Planes
from tools.bench_p3 import timeit
dev = torch.device("cuda")
lv = Levels([(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)], 4)
g = ConvGeom(lv, 256, 256, 3, 1, 1); g.x3 = True
mode = sys.argv[1] if len(sys.argv) > 1 else "rand"
x = torch.relu(torch.randn(lv.rows, 256, device=dev)); w = torch.randn(256 * 9, 256, device=dev) * 0.05
if mode == "zero": x.zero_(); w.zero_()
if mode == "dense": x = torch.randn(lv.rows, 256, device=dev)
xp, wp = Planes.from_float(x), Planes.from_float(w)
y1, y2 = torch.empty(lv.rows, 256, device=dev), torch.empty(lv.rows, 256, device=dev)
flop = 2.0 * lv.rows * 256 * 256 * 9 * 2
for t, fl in [(1, 0), (1, K.STAGES3), (2, 0), (5, 0), (5, K.STAGES3), (6, 0), (3, 0), (3, K.STAGES3)]:
    us = timeit(lambda: K.conv_fwd_pair(g, dict(x=xp, w=wp, y=y1), dict(x=xp, w=wp, y=y2), tile=t | fl))
    print(f"dbg={os.environ.get('RADET_DBG_IGEMM','0')} data={mode} pair tile {t} fl {fl:#x}: {us:8.1f} us {flop/us/1e6:7.1f} TFLOP/s", flush=True)
us = timeit(lambda: K.conv_fwd_pair(g, dict(x=x, w=w.view(-1), y=y1), dict(x=x, w=w.view(-1), y=y2), tile=0x201))
print(f"   X3 pair tile 1 data={mode}: {us:8.1f} us {flop/us/1e6:7.1f} TFLOP/s")
