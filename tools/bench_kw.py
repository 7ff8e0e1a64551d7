#!/usr/bin/env python
"""The K-divided 64 x 64 implicit-GEMM tile (tile 7: every wave accumulates the whole tile over its own 16 channels of a
64-channel K step) against the 2 x 2-wave tiles on the backbone / neck shapes of the headline step.  GPU only.

    python tools/bench_kw.py [fwd|dgrad]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radet_amd import kernels as K  # noqa: E402
from radet_amd.kernels import ConvGeom, Levels  # noqa: E402
from tools.bench_p3 import timeit, rel  # noqa: E402

SHAPES = [  # name, (H, W) of the input, cin, cout, k, stride
    ("layer1 3x3", (120, 160), 64, 64, 3, 1), ("layer1 up", (120, 160), 64, 256, 1, 1), ("layer1 down", (120, 160), 256, 64, 1, 1),
    ("layer2 3x3", (60, 80), 128, 128, 3, 1), ("layer2 up", (60, 80), 128, 512, 1, 1), ("layer2 down", (60, 80), 512, 128, 1, 1),
    ("layer2 3x3 s2", (120, 160), 128, 128, 3, 2),
    ("layer3 3x3", (30, 40), 256, 256, 3, 1), ("layer3 up", (30, 40), 256, 1024, 1, 1), ("layer3 down", (30, 40), 1024, 256, 1, 1),
    ("layer4 3x3", (15, 20), 512, 512, 3, 1), ("layer4 up", (15, 20), 512, 2048, 1, 1), ("layer4 down", (15, 20), 2048, 512, 1, 1),
    ("fpn out P3", (60, 80), 256, 256, 3, 1), ("fpn lateral C4", (30, 40), 1024, 256, 1, 1),
]


def main(what="fwd"):
    dev = torch.device("cuda")
    B = 4
    tot_ref = tot_kw = 0.0
    for name, hw, cin, cout, k, stride in SHAPES:
        lv = Levels([hw], B)
        g = ConvGeom(lv, cin, cout, k, stride, k // 2)
        g.x3 = True
        torch.manual_seed(0)
        x = torch.relu(torch.randn(lv.rows, cin, device=dev))
        w = torch.randn(cout * k * k * cin, device=dev) * 0.05
        y0 = torch.empty(g.lout.rows, cout, device=dev)
        y1 = torch.empty_like(y0)
        nk = k * k * cin // 64
        sks = [sk for sk in (1, 2, 3, 4, 6, 8) if sk == 1 or nk // sk >= 2]

        def run(t, y):
            K.conv_fwd(g, x, w, None, y, relu=True, tile=t)
        ref_c = [t | 0x200 | (sk << 12) for t in (1, 2, 3) for sk in (0, 1, 2, 3, 4, 6, 8) if sk <= 1 or (k * k * cin // 32) // sk >= 4]
        t_ref, best_ref = min((timeit(lambda: run(t, y0), n=10, warm=2), t) for t in ref_c)
        run(3 | 0x200, y0)
        res = []
        per = {}
        for tid in (7, 8):
            if cin % (64 if tid == 7 else 32):
                continue
            nkt = k * k * cin // (64 if tid == 7 else 32)
            for sk in [sk for sk in (1, 2, 3, 4, 6, 8) if sk == 1 or nkt // sk >= 3]:
                t = tid | (sk << 12)
                y1.zero_()
                run(t, y1)
                torch.cuda.synchronize()
                us = timeit(lambda: run(t, y1), n=10, warm=2)
                res.append((us, (tid, sk), rel(y1, y0)))
                per[tid] = min(per.get(tid, 1e9), us)
        t_kw, sk_kw, err = min(res)
        worst = max(r[2] for r in res)
        flop = 2.0 * g.lout.rows * cout * cin * k * k
        tot_ref += t_ref
        tot_kw += min(t_kw, t_ref)
        print(f"{name:16s} M={g.lout.rows:6d} {cin:4d}->{cout:4d} k{k}s{stride}: best 2x2 tile {best_ref:#9x} {t_ref:6.1f} us "
              f"{flop / t_ref / 1e6:6.1f} TF | K-divided (tile, sk)={sk_kw} {t_kw:6.1f} us {flop / t_kw / 1e6:6.1f} TF ({t_ref / t_kw:4.2f}x) "
              f"max rel diff {worst:.1e} | " + " ".join(f"t{t}:{u:.1f}" for t, u in sorted(per.items())), flush=True)
    print(f"sum: {tot_ref:.0f} us -> {tot_kw:.0f} us with the better of the two per shape")


def wgrad():
    dev = torch.device("cuda")
    B = 4
    tot_ref = tot_kw = 0.0
    for name, hw, cin, cout, k, stride in SHAPES:
        if cin <= 64 or cout <= 64:
            continue
        lv = Levels([hw], B)
        g = ConvGeom(lv, cin, cout, k, stride, k // 2)
        g.x3 = True
        M, kk = g.lout.rows, k * k
        torch.manual_seed(0)
        x = torch.relu(torch.randn(lv.rows, cin, device=dev))
        dy = torch.randn(M, cout, device=dev) * 0.01
        ref = None
        per = {}
        worst = 0.0
        for tname, fl, t, tn in (("128x128", 1 << 4, 128, 128), ("64x64", 2 << 4, 64, 64), ("64x64/32px", 2 << 4 | 0x80, 64, 64),
                                 ("128x64", 3 << 4, 128, 64), ("64x64 KD4", 2 << 4 | 0x400, 64, 64), ("64x64 KD2", 2 << 4 | 0x800, 64, 64)):
            tiles = -(-cout // t) * -(-cin // tn) * kk
            for blocks in (256, 512, 768, 1024):
                S = max(1, min(64, round(blocks / tiles), (M + 127) // 128))
                g.wgrad_flags, g.nsplit = fl | 0x40, S
                slabs = torch.zeros(S * cout * kk * cin, device=dev)
                K.conv_wgrad(g, dy, x, slabs)
                torch.cuda.synchronize()
                tot = slabs.view(S, -1).sum(0)
                if ref is None:
                    ref = tot
                worst = max(worst, rel(tot, ref))
                us = timeit(lambda: K.conv_wgrad(g, dy, x, slabs), n=10, warm=2) + S * cout * kk * cin * 8 / 3e12 * 1e6
                if us < per.get(tname, (1e9,))[0]:
                    per[tname] = (us, S)
        t_ref = min(v[0] for n_, v in per.items() if "KD" not in n_)
        t_kw = min(v[0] for n_, v in per.items() if "KD" in n_)
        tot_ref += t_ref
        tot_kw += min(t_ref, t_kw)
        flop = 2.0 * M * cout * cin * kk
        print(f"{name:16s} M={M:6d} {cin:4d}->{cout:4d} k{k}s{stride}: best old {t_ref:6.1f} us {flop / t_ref / 1e6:6.1f} TF | pixel-divided "
              f"{t_kw:6.1f} us ({t_ref / t_kw:4.2f}x) max rel diff {worst:.1e} | " + " ".join(f"{n_}:{v[0]:.1f}(S{v[1]})" for n_, v in per.items()),
              flush=True)
    print(f"sum (incl. slab cost): {tot_ref:.0f} us -> {tot_kw:.0f} us with the better of the two per shape")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "wgrad":
        wgrad()
        sys.exit(0)
    main(sys.argv[1] if len(sys.argv) > 1 else "fwd")
