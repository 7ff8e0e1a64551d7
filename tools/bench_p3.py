#!/usr/bin/env python
"""Plane-operand conv GEMMs (radet_conv2d_igemm +0x2000000, radet_conv2d_wgrad flags 0x200) against the in-register split
(X3) kernels: agreement on the same data and time per launch over the tile configurations.  GPU only.

    python tools/bench_p3.py [tower|layer3|all]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radet_amd import kernels as K  # noqa: E402
from radet_amd.kernels import ConvGeom, Levels, Planes  # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / n * 1e3      # us


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def run_shape(name, lv, cin, cout, k, stride, pair=False, tiles=None):
    dev = torch.device("cuda")
    g = ConvGeom(lv, cin, cout, k, stride, k // 2)
    g.x3 = True
    torch.manual_seed(0)
    x = torch.relu(torch.randn(lv.rows, cin, device=dev))
    w = torch.randn(cout * k * k, cin, device=dev) * 0.05
    y0 = torch.empty(g.lout.rows, cout, device=dev)
    xp, wp = Planes.from_float(x), Planes.from_float(w)
    assert torch.equal(xp.to_float(), x), "plane split is not exact"
    flop = 2.0 * g.lout.rows * cout * cin * k * k * (2 if pair else 1)
    x2 = torch.relu(torch.randn(lv.rows, cin, device=dev))
    x2p = Planes.from_float(x2)
    y0b = torch.empty_like(y0)

    def ref(t=0x200 | 1):
        if pair:
            K.conv_fwd_pair(g, dict(x=x, w=w.view(-1), y=y0), dict(x=x2, w=w.view(-1), y=y0b), tile=t)
        else:
            K.conv_fwd(g, x, w.view(-1), None, y0, tile=t)
    ref()
    t_ref = min(timeit(lambda: ref(0x200 | t)) for t in (1, 2, 3))
    print(f"== {name}: M={g.lout.rows} {cin}->{cout} k{k}s{stride} pair={pair}  X3 (in-register split, best tile): "
          f"{t_ref:8.1f} us  {flop / t_ref / 1e6:7.1f} TFLOP/s")
    y1 = torch.empty_like(y0)
    y1b = torch.empty_like(y0)
    cands = tiles or [(1, 0), (2, 0), (3, 0), (3, K.STAGES3), (5, 0), (6, 0)]
    for t, fl in cands:
        def run():
            if pair:
                K.conv_fwd_pair(g, dict(x=xp, w=wp, y=y1), dict(x=x2p, w=wp, y=y1b), tile=t | fl)
            else:
                K.conv_fwd(g, xp, wp, None, y1, tile=t | fl)
        y1.zero_()
        run()
        torch.cuda.synchronize()
        err = rel(y1, y0)
        us = timeit(run)
        print(f"   P3 tile {t} flags {fl:#010x}: {us:8.1f} us  {flop / us / 1e6:7.1f} TFLOP/s  ({t_ref / us:4.2f}x)  max rel diff vs X3 {err:.2e}")
    return g, x, xp


def run_wgrad(lv, c=256):
    dev = torch.device("cuda")
    g = ConvGeom(lv, c, c, 3, 1, 1)
    g.x3 = True
    M = g.lout.rows
    torch.manual_seed(1)
    x = torch.relu(torch.randn(lv.rows, c, device=dev))
    dy = torch.randn(M, c, device=dev) * 0.01
    slabs0 = torch.empty(g.nsplit * c * 9 * c, device=dev)
    slabs1 = torch.empty_like(slabs0)
    K.conv_wgrad(g, dy, x, slabs0)
    xp, dyp = Planes.from_float(x), Planes.from_float(dy)
    K.conv_wgrad(g, dyp, xp, slabs1)
    torch.cuda.synchronize()
    s0 = slabs0.view(g.nsplit, -1).sum(0)
    s1 = slabs1.view(g.nsplit, -1).sum(0)
    flop = 2.0 * M * c * c * 9
    t0 = timeit(lambda: K.conv_wgrad(g, dy, x, slabs0))
    t1 = timeit(lambda: K.conv_wgrad(g, dyp, xp, slabs1))
    print(f"== wgrad9 M={M} {c}->{c} S={g.nsplit}: X3 {t0:8.1f} us {flop / t0 / 1e6:7.1f} TFLOP/s | P3 {t1:8.1f} us "
          f"{flop / t1 / 1e6:7.1f} TFLOP/s ({t0 / t1:4.2f}x)  max rel diff {rel(s1, s0):.2e}")
    # reference in fp64 on a sub-problem is done by the pytest suite; here agreement with the X3 kernel


def run_gn(lv):
    dev = torch.device("cuda")
    R = lv.rows
    torch.manual_seed(2)
    z = torch.randn(R, 256, device=dev)
    gam, bet = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev) * 0.1
    y, yp = torch.empty_like(z), Planes(R, 256, device=dev)
    stats = torch.empty(len(lv) * lv.B * 64, device=dev)
    ws = torch.empty(K.gn_ws_floats(lv), device=dev)
    K.gn_relu_fwd(lv, z, gam, bet, y, stats, ws)
    y2 = torch.empty_like(z)
    K.gn_relu_fwd_p(lv, z, gam, bet, y2, yp, stats, ws)
    torch.cuda.synchronize()
    print("== GN fwd planes exact:", torch.equal(yp.to_float(), y), torch.equal(y2, y),
          f"fp32 {timeit(lambda: K.gn_relu_fwd(lv, z, gam, bet, y, stats, ws)):.1f} us, planes only "
          f"{timeit(lambda: K.gn_relu_fwd_p(lv, z, gam, bet, None, yp, stats, ws)):.1f} us")
    dy = torch.randn(R, 256, device=dev)
    dz, dzp = torch.empty_like(z), Planes(R, 256, device=dev)
    dg, db = torch.empty(256, device=dev), torch.empty(256, device=dev)
    K.gn_relu_bwd(lv, dy, z, stats, gam, bet, dz, dg, db, ws)
    K.gn_relu_bwd_p(lv, dy, z, stats, gam, bet, None, dzp, dg, db, ws)
    torch.cuda.synchronize()
    print("== GN bwd planes exact:", torch.equal(dzp.to_float(), dz),
          f"fp32 {timeit(lambda: K.gn_relu_bwd(lv, dy, z, stats, gam, bet, dz, dg, db, ws)):.1f} us, planes "
          f"{timeit(lambda: K.gn_relu_bwd_p(lv, dy, z, stats, gam, bet, None, dzp, dg, db, ws)):.1f} us")
    x = torch.randn(R, 256, device=dev)
    xp = Planes(R, 256, device=dev)
    print(f"== split_planes R={R}: {timeit(lambda: K.split_planes(x, xp)):.1f} us")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    B = 4
    plv = Levels([(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)], B)
    if what in ("tower", "all"):
        run_gn(plv)
        run_shape("tower", plv, 256, 256, 3, 1, pair=False)
        run_shape("tower pair", plv, 256, 256, 3, 1, pair=True, tiles=[(1, 0), (2, 0), (3, 0), (5, 0), (6, 0)])
        run_wgrad(plv)
    if what in ("layer3", "all"):
        run_shape("layer3 3x3", Levels([(30, 40)], B), 256, 256, 3, 1)
        run_shape("layer3 1x1 up", Levels([(30, 40)], B), 256, 1024, 1, 1)
        run_shape("layer3 1x1 down", Levels([(30, 40)], B), 1024, 256, 1, 1)
        run_shape("layer2 3x3", Levels([(60, 80)], B), 128, 128, 3, 1)
        run_shape("layer2 1x1 up", Levels([(60, 80)], B), 128, 512, 1, 1)
        run_shape("fpn out P3", Levels([(60, 80)], B), 256, 256, 3, 1)
