#!/usr/bin/env python
"""The 128 x 128 split-at-fill implicit-GEMM tile (tile 9, conv_igemm_sf_kernel: operands split into bf16 planes once per
workgroup on the way into LDS) against the tiles the tuner chooses from today (2 x 2-wave register split 1-3, K-divided 7 / 8)
on the backbone / neck shapes of the headline step.  GPU only.     python tools/bench_sf.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radet_amd import kernels as K  # noqa: E402
from radet_amd.kernels import ConvGeom, Levels  # noqa: E402
from tools.bench_p3 import timeit, rel  # noqa: E402
from tools.bench_kw import SHAPES  # noqa: E402


def main(B=4):
    dev = torch.device("cuda")
    tot_ref = tot_new = 0.0
    for name, hw, cin, cout, k, stride in SHAPES:
        lv = Levels([hw], B)
        g = ConvGeom(lv, cin, cout, k, stride, k // 2)
        g.x3 = True
        torch.manual_seed(0)
        x = torch.relu(torch.randn(lv.rows, cin, device=dev))
        w = torch.randn(cout * k * k * cin, device=dev) * 0.05
        bias = torch.randn(cout, device=dev)
        y0 = torch.empty(g.lout.rows, cout, device=dev)
        y1 = torch.empty_like(y0)
        add = torch.randn_like(y0)

        def run(t, y):
            K.conv_fwd(g, x, w, bias, y, addend=add, relu=True, tile=t)
        nk32 = k * k * cin // 32
        cands = [t | 0x200 | (sk << 12) for t in (1, 2, 3) for sk in (0, 1, 2, 3, 4, 6, 8) if sk <= 1 or nk32 // sk >= 4]
        for tid in (7, 8):
            if cin % (64 if tid == 7 else 32) == 0:
                nkt = k * k * cin // (64 if tid == 7 else 32)
                cands += [tid | (sk << 12) for sk in (1, 2, 3, 4, 6, 8) if sk == 1 or nkt // sk >= 3]
        t_ref, best_ref = min((timeit(lambda: run(t, y0), n=10, warm=2), t) for t in cands)
        run(3 | 0x200, y0)
        res = []
        for sk in [sk for sk in (1, 2, 3, 4, 6, 8, 12) if sk == 1 or nk32 // sk >= 3]:
            t = 9 | (sk << 12)
            y1.zero_()
            run(t, y1)
            torch.cuda.synchronize()
            res.append((timeit(lambda: run(t, y1), n=10, warm=2), sk, rel(y1, y0)))
        t_new, sk_new, _ = min(res)
        worst = max(r[2] for r in res)
        flop = 2.0 * g.lout.rows * cout * cin * k * k
        tot_ref += t_ref
        tot_new += min(t_new, t_ref)
        print(f"{name:16s} M={g.lout.rows:6d} {cin:4d}->{cout:4d} k{k}s{stride}: best today {best_ref:#9x} {t_ref:6.1f} us "
              f"{flop / t_ref / 1e6:6.1f} TF | split-at-fill sk={sk_new} {t_new:6.1f} us {flop / t_new / 1e6:6.1f} TF ({t_ref / t_new:4.2f}x) "
              f"max rel diff {worst:.1e} | " + " ".join(f"sk{r[1]}:{r[0]:.1f}" for r in res), flush=True)
    print(f"sum: {tot_ref:.0f} us -> {tot_new:.0f} us with the better of the two per shape")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
