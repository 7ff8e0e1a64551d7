"""Predictor head convs (256 -> 21 / 4 / 1, 3x3) on the headline pyramid: implicit-GEMM launches (tuned tiles) against the
direct convolution from an LDS patch (radet_pred3x3_patch).  python tools/bench_pred.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radet_amd import kernels as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lv = K.Levels([(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)], B)
x = torch.randn(lv.rows, 256, device="cuda")
heads = []
for c in (21, 4, 1):
    g = K.ConvGeom(lv, 256, c, 3, 1, 1)
    g.x3 = True
    K.autotune(g, need_dgrad=False)
    heads.append((g, torch.randn(c, 9, 256, device="cuda") * 0.02, torch.randn(c, device="cuda"),
                  torch.empty(lv.rows, c, device="cuda"), c))


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n * 1e3


def igemm():
    for g, w, b, y, c in heads:
        K.conv_fwd(g, x, w, b, y)


def patch():
    K.pred_conv_patch(lv, x, heads[0][1:])
    K.pred_conv_patch(lv, x, heads[1][1:], heads[2][1:])


print(f"B={B} rows={lv.rows}: implicit GEMM x3 launches {timeit(igemm):.1f} us, LDS patch x2 launches {timeit(patch):.1f} us "
      f"(cls alone {timeit(lambda: K.pred_conv_patch(lv, x, heads[0][1:])):.1f} us)")
