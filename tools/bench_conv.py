"""Micro-benchmark of the implicit-GEMM conv kernels (tile configs x shapes). GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radet_amd import kernels as K


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


def main():
    dev = "cuda"
    shapes = [  # name, levels(hw), B, cin, cout, k, stride
        ("tower 5lvl B4", [(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)], 4, 256, 256, 3, 1),
        ("tower big M=131072", [(128, 256)], 4, 256, 256, 3, 1),
        ("l2 1x1 512->128", [(60, 80)], 4, 512, 128, 1, 1),
        ("l2 1x1 128->512", [(60, 80)], 4, 128, 512, 1, 1),
        ("l2 3x3 128", [(60, 80)], 4, 128, 128, 3, 1),
        ("l3 1x1 1024->256", [(30, 40)], 4, 1024, 256, 1, 1),
        ("l3 1x1 256->1024", [(30, 40)], 4, 256, 1024, 1, 1),
        ("l3 3x3 256", [(30, 40)], 4, 256, 256, 3, 1),
        ("l4 1x1 2048->512", [(15, 20)], 4, 2048, 512, 1, 1),
        ("l4 3x3 512", [(15, 20)], 4, 512, 512, 3, 1),
        ("l1 3x3 64", [(120, 160)], 4, 64, 64, 3, 1),
        ("l1 1x1 64->256", [(120, 160)], 4, 64, 256, 1, 1),
    ]
    for name, hw, B, cin, cout, k, s in shapes:
        lv = K.Levels(hw, B)
        g = K.ConvGeom(lv, cin, cout, k, s, k // 2)
        x = torch.randn(lv.rows, cin, device=dev)
        w = torch.randn(cout, k * k, cin, device=dev) * 0.05
        y = torch.empty(g.lout.rows, cout, device=dev)
        fl = 2.0 * g.lout.rows * cout * cin * k * k
        res = []
        for tile in (1, 2, 3, 0, 0x201, 0x202, 0x203):
            t = timeit(lambda: K.conv_fwd(g, x, w, None, y, relu=True, tile=tile))
            res.append(f"t{tile:x}:{fl / t / 1e12:5.1f}")
        dy = torch.randn(g.lout.rows, cout, device=dev)
        slabs = torch.empty(g.nsplit * cout * k * k * cin, device=dev)
        t = timeit(lambda: K.conv_wgrad(g, dy, x, slabs, None))
        print(f"{name:22s} M={g.lout.rows:7d} " + " ".join(res) + f" | wgrad S={g.nsplit:2d}: {fl / t / 1e12:6.1f}TF")


if __name__ == "__main__":
    main()
