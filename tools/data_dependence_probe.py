import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radet_amd import kernels as K
B = 4
lv = K.Levels([(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)], B)
g = K.ConvGeom(lv, 256, 256, 3, 1, 1); g.x3 = True
x = [torch.randn(lv.rows, 256, device="cuda") for _ in range(2)]
w = [torch.randn(256 * 9 * 256, device="cuda") * 0.02 for _ in range(2)]
y = [torch.empty(lv.rows, 256, device="cuda") for _ in range(2)]
fl = 2 * 2.0 * lv.rows * 256 * 256 * 9
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n * 1e3
for name, fx in (("gaussian activations", lambda t: t), ("ReLU'd activations (half zeros)", torch.relu), ("all-zero activations", torch.zeros_like),
                 ("constant activations", torch.ones_like)):
    xs = [fx(t) for t in x]
    fn = lambda: K.conv_fwd_pair(g, dict(x=xs[0], w=w[0], y=y[0]), dict(x=xs[1], w=w[1], y=y[1]), relu=True, tile=0x201)
    us = t(fn)
    print(f"128x128 BK32, {name}: {us:7.1f} us  {fl / us / 1e6:6.1f} TF fp32-equivalent")
g.x3 = False
for name, fx in (("gaussian activations", lambda t: t), ("ReLU'd activations (half zeros)", torch.relu)):
    xs = [fx(t) for t in x]
    fn = lambda: K.conv_fwd_pair(g, dict(x=xs[0], w=w[0], y=y[0]), dict(x=xs[1], w=w[1], y=y[1]), relu=True, tile=0x202)
    us = t(fn)
    print(f"native fp32 MFMA 128x64 BK32, {name}: {us:7.1f} us  {fl / us / 1e6:6.1f} TF")
