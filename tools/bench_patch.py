#!/usr/bin/env python
"""The head towers' 3x3 conv from an LDS patch (radet_conv3x3_patch_p) against the plane-operand implicit GEMM (256 x 128
tile): results, time alone, and time of two launches on two streams (the two tower chains).  python tools/bench_patch.py [B]"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radet_amd import kernels as K  # noqa: E402
from radet_amd.kernels import ConvGeom, Levels, Planes  # noqa: E402
from tools.bench_p3 import timeit  # noqa: E402


def planes_of(t):
    p = Planes(t.shape[0], t.shape[1], device=t.device)
    K.split_planes(t, p)
    return p


def main(B=4, C=256):
    dev = torch.device("cuda")
    lv = Levels([(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)], B)
    g = ConvGeom(lv, C, C, 3, 1, 1)
    torch.manual_seed(0)
    x = torch.relu(torch.randn(lv.rows, C, device=dev))
    w = torch.randn(C * 9, C, device=dev) * 0.05            # rows (n, tap) x Cin  == OHWI
    bias = torch.randn(C, device=dev)
    xp, wp = planes_of(x), planes_of(w)
    y0 = torch.empty(lv.rows, C, device=dev)
    y1 = torch.zeros_like(y0)
    t6 = 6 | (1 << 12)
    K.conv_fwd(g, xp, wp, bias, y0, tile=t6)
    K.conv3x3_patch(lv, xp, wp, bias, y1, C, C)
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d  # noqa: F841  (not used: the fp64 reference below is plain matmul per tap)
    # fp64 reference on a sample of rows of every level
    err0 = err1 = 0.0
    xd, wd = x.double(), w.double().view(C, 9, C)
    for l, (h, wd_) in enumerate(lv.hw):
        off = lv.offsets[l]
        for n in (0, B - 1):
            for (yy, xx) in ((0, 0), (h - 1, wd_ - 1), (h // 2, wd_ // 3), (0, wd_ - 1)):
                acc = bias.double().clone()
                for t in range(9):
                    iy, ix = yy + t // 3 - 1, xx + t % 3 - 1
                    if 0 <= iy < h and 0 <= ix < wd_:
                        acc += wd[:, t, :] @ xd[off + n * h * wd_ + iy * wd_ + ix]
                row = off + n * h * wd_ + yy * wd_ + xx
                sc = float(acc.abs().max())
                err0 = max(err0, float((y0[row].double() - acc).abs().max()) / sc)
                err1 = max(err1, float((y1[row].double() - acc).abs().max()) / sc)
    print(f"B={B}: max rel error vs fp64 on sampled rows: implicit GEMM {err0:.2e}, patch {err1:.2e}; "
          f"patch vs implicit GEMM max abs diff {float((y0 - y1).abs().max()):.2e} (max |y| {float(y0.abs().max()):.1f})")
    # dgrad orientation: flip, transposed weight planes, addend
    wt = w.view(C, 9, C).permute(2, 1, 0).reshape(C * 9, C).contiguous()     # rows (c, tap) x Cout
    wtp = planes_of(wt)
    dy = torch.randn(lv.rows, C, device=dev)
    dyp = planes_of(dy)
    add = torch.randn(lv.rows, C, device=dev)
    d0, d1 = torch.empty_like(y0), torch.zeros_like(y0)
    K.conv_dgrad(g, dyp, wtp, d0, addend=add, tile=t6)
    K.conv3x3_patch(lv, dyp, wtp, None, d1, C, C, addend=add, flip=True)
    torch.cuda.synchronize()
    print(f"dgrad: patch vs implicit GEMM max abs diff {float((d0 - d1).abs().max()):.2e} (max |dx| {float(d0.abs().max()):.1f})")
    fl = 2.0 * lv.rows * C * C * 9
    a = timeit(lambda: K.conv_fwd(g, xp, wp, bias, y0, tile=t6), n=20, warm=3)
    b = timeit(lambda: K.conv3x3_patch(lv, xp, wp, bias, y1, C, C), n=20, warm=3)
    print(f"alone: implicit GEMM 256x128 {a:.1f} us ({fl / a / 1e6:.0f} TF) | patch {b:.1f} us ({fl / b / 1e6:.0f} TF)  {a / b:.2f}x")
    side = torch.cuda.Stream()
    y2, y3 = torch.empty_like(y0), torch.empty_like(y0)

    def two(fn_a, fn_b):
        ev = torch.cuda.Event(); ev.record()
        fn_a()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            fn_b()
            e2 = torch.cuda.Event(); e2.record()
        torch.cuda.current_stream().wait_event(e2)
    a2 = timeit(lambda: two(lambda: K.conv_fwd(g, xp, wp, bias, y0, tile=t6), lambda: K.conv_fwd(g, xp, wp, bias, y2, tile=t6)), n=20, warm=3)
    b2 = timeit(lambda: two(lambda: K.conv3x3_patch(lv, xp, wp, bias, y1, C, C), lambda: K.conv3x3_patch(lv, xp, wp, bias, y3, C, C)), n=20, warm=3)
    print(f"two chains: implicit GEMM {a2:.1f} us per pair ({2 * fl / a2 / 1e6:.0f} TF) | patch {b2:.1f} us ({2 * fl / b2 / 1e6:.0f} TF)  {a2 / b2:.2f}x")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
