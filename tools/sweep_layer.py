"""Time every (tile, K step, split-K, stages) candidate of the implicit-GEMM forward kernel on one layer shape:
    sweep_layer.py cin cout k H W [B]        -> one line per candidate, sorted by time"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radet_amd import kernels as K
cin, cout, k, H, W = (int(v) for v in sys.argv[1:6])
B = int(sys.argv[6]) if len(sys.argv) > 6 else 4
lv = K.Levels([(H, W)], B)
g = K.ConvGeom(lv, cin, cout, k, 1, k // 2)
x = torch.randn(lv.rows, cin, device="cuda")
w = torch.randn(cout * k * k * cin, device="cuda") * 0.05
y = torch.empty(lv.rows, cout, device="cuda")
cands = []
for t in (1, 2, 3, 4):
    for bk in (0, 0x200):
        for st in (0, K.STAGES3):
            for sk in (0, 1, 2, 3, 4, 5, 6, 8):
                cands.append(t | bk | st | (sk << 12))
        if t != 1:
            cands += [t | bk | (w * K.STREAMK) for w in (1, 2, 3, 4)]
res = []
flops = 2.0 * lv.rows * cin * cout * k * k
for t in cands:
    try:
        K.conv_fwd(g, x, w, None, y, relu=True, tile=t)
    except Exception as e:
        continue
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); K.conv_fwd(g, x, w, None, y, relu=True, tile=t); e.record(); e.synchronize()
        best = min(best, s.elapsed_time(e))
    res.append((best, t))
names = {1: "128x128", 2: "128x64", 3: "64x64", 4: "128x32"}
for ms, t in sorted(res)[:14]:
    print(f"{names[t & 0xff]:8s} BK{32 if t & 0x200 else 16} stages{3 if t & K.STAGES3 else 2} sk={(t >> 12) & 15} streamk={(t >> 20) & 7}: "
          f"{ms * 1e3:7.1f} us  {flops / ms / 1e9:6.1f} TF")
