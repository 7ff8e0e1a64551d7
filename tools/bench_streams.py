"""Do two independent tower GEMMs overlap usefully on two HIP streams?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radet_amd import kernels as K
lv = K.Levels([(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)], 4)
g = K.ConvGeom(lv, 256, 256, 3, 1, 1)
mk = lambda: (torch.randn(lv.rows, 256, device="cuda"), torch.randn(256, 9, 256, device="cuda") * 0.05, torch.empty(lv.rows, 256, device="cuda"))
xa, wa, ya = mk(); xb, wb, yb = mk()
dy = torch.randn(lv.rows, 256, device="cuda")
slabs = torch.empty(g.nsplit * 256 * 9 * 256, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
TILE = 0x203
def seq(n):
    for _ in range(n):
        K.conv_fwd(g, xa, wa, None, ya, tile=TILE, splitk=False); K.conv_fwd(g, xb, wb, None, yb, tile=TILE, splitk=False)
def par(n):
    for _ in range(n):
        with torch.cuda.stream(s1): K.conv_fwd(g, xa, wa, None, ya, tile=TILE, splitk=False)
        with torch.cuda.stream(s2): K.conv_fwd(g, xb, wb, None, yb, tile=TILE, splitk=False)
def seq_w(n):
    for _ in range(n):
        K.conv_fwd(g, xa, wa, None, ya, tile=TILE, splitk=False); K.conv_wgrad(g, dy, xb, slabs, None)
def par_w(n):
    for _ in range(n):
        with torch.cuda.stream(s1): K.conv_fwd(g, xa, wa, None, ya, tile=TILE, splitk=False)
        with torch.cuda.stream(s2): K.conv_wgrad(g, dy, xb, slabs, None)
import time
for name, fn in (("seq fwd+fwd", seq), ("par fwd+fwd", par), ("seq fwd+wgrad", seq_w), ("par fwd+wgrad", par_w)):
    fn(3); torch.cuda.synchronize()
    t = time.perf_counter(); fn(20); torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print(f"{name:16s} {dt*1e6:8.1f} us per pair  -> {2*2*lv.rows*256*2304/dt/1e12:6.1f} TF")
