"""fp16 hi / lo arithmetic (h2) against the bf16-plane arithmetic (x3) per conv shape and tile: time per launch alone on the
device, with and without output amax tracking.   python tools/bench_h2.py [fwd|wgrad]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from radet_amd import kernels as K

SHAPES = [  # B, Cin, Cout, H, W, k, stride
    (4, 64, 64, 120, 160, 1, 1), (4, 64, 64, 120, 160, 3, 1), (4, 64, 256, 120, 160, 1, 1), (4, 256, 64, 120, 160, 1, 1),
    (4, 256, 128, 120, 160, 1, 1), (4, 128, 128, 60, 80, 3, 1), (4, 128, 512, 60, 80, 1, 1), (4, 512, 128, 60, 80, 1, 1),
    (4, 256, 256, 30, 40, 3, 1), (4, 256, 1024, 30, 40, 1, 1), (4, 1024, 256, 30, 40, 1, 1),
    (4, 512, 512, 15, 20, 3, 1), (4, 512, 2048, 15, 20, 1, 1), (4, 2048, 512, 15, 20, 1, 1), (4, 256, 256, 60, 80, 3, 1),
]


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        e.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
    dev = "cuda"
    for (B, Cin, Cout, H, W, k, s) in SHAPES:
        lv = K.Levels([(H, W)], B)
        g = K.ConvGeom(lv, Cin, Cout, k, s, k // 2)
        x = torch.randn(lv.rows, Cin, device=dev).clamp_min(0)
        w = torch.randn(Cout * k * k * Cin, device=dev) * 0.05
        y = torch.empty(g.lout.rows, Cout, device=dev)
        slots = [K.new_amax(dev) for _ in range(3)]
        K.absmax(x, slots[0]); K.absmax(w, slots[1])
        keys = [K.register_amax(x, slots[0]), K.register_amax(w, slots[1])]
        row = [f"M={g.lout.rows:6d} {Cin:4d}->{Cout:4d} k{k}"]
        if what == "fwd":
            for tile in (1, 2, 3, 7, 8, 8 | 0x20000, 8 | 0x40000):
                if (tile & 0xFF) == 7 and Cin % 64:
                    row.append("      -      ")
                    continue
                g.x3, g.h2 = True, False
                t3 = timeit(lambda: K.conv_fwd(g, x, w, None, y, relu=True, tile=tile))
                g.h2 = True
                th = timeit(lambda: K.conv_fwd(g, x, w, None, y, relu=True, tile=tile))
                ky = K.register_amax(y, slots[2])
                K._SCALES.clear()
                tha = timeit(lambda: K.conv_fwd(g, x, w, None, y, relu=True, tile=tile))
                dummy = K.new_amax(dev)

                def zf(slot):
                    slot.zero_()                     # (a fresh slot per launch, as in the step: every wave of the first round raises it)
                    K.conv_fwd(g, x, w, None, y, relu=True, tile=tile)
                thd = timeit(lambda: zf(dummy))
                thz = timeit(lambda: zf(slots[2]))
                K.unregister_amax([ky])
                row.append(f"t{tile & 0xFF}{chr(97 + (tile >> 17))}: {t3:5.1f} {th:5.1f} {tha:5.1f} [{thd:5.1f} {thz:5.1f}]")
            xq, wq = K.Planes.from_float(x, kind="h2"), K.Planes.from_float(w.view(Cout * k * k, Cin), kind="h2")
            for tile in (1, 2, 3, 3 | 0x20000):
                tq = timeit(lambda: K.conv_fwd(g, xq, wq, None, y, relu=True, tile=tile))
                row.append(f"pairs t{tile & 0xFF}{chr(97 + (tile >> 17))}: {tq:5.1f}")
        else:
            dy = torch.randn(g.lout.rows, Cout, device=dev)
            K.absmax(dy, slots[2])
            keys.append(K.register_amax(dy, slots[2]))
            for fl in ((1 << 4) | 0x40, (2 << 4) | 0x40, (3 << 4) | 0x40, (2 << 4) | 0x40 | 0x400, (2 << 4) | 0x40 | 0x800):
                g.wgrad_flags, g.nsplit = fl, max(1, min(16, 512 // (max(1, Cout // 64) * max(1, Cin // 64) * k * k)))
                slabs = torch.empty(g.nsplit * Cout * k * k * Cin, device=dev)
                g.x3, g.h2 = True, False
                t3 = timeit(lambda: K.conv_wgrad(g, dy, x, slabs))
                g.h2 = True
                th = timeit(lambda: K.conv_wgrad(g, dy, x, slabs))
                row.append(f"{fl:#5x}: {t3:5.1f} {th:5.1f}")
        K.unregister_amax(keys)
        print("  ".join(row), flush=True)


if __name__ == "__main__":
    print("us per launch: bf16x3  fp16x2  fp16x2 + output amax [+ a fill kernel per launch: of a dummy / of the output slot]" if (len(sys.argv) < 2 or sys.argv[1] == "fwd") else "us per launch: bf16x3  fp16x2")
    main()
