"""fp32 products from three bf16 planes (tile flag X3) against the native fp32 MFMA kernel: error vs an fp64 convolution
and time, per tile, on the tower shape and two backbone shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from radet_amd import kernels as K

def run(cin, cout, k, H, W, B=4):
    lv = K.Levels([(H, W)], B)
    g = K.ConvGeom(lv, cin, cout, k, 1, k // 2)
    gen = torch.Generator().manual_seed(0)
    x4 = torch.randn(B, cin, H, W, generator=gen)
    w4 = torch.randn(cout, cin, k, k, generator=gen) / (cin * k * k) ** 0.5
    ref = F.conv2d(x4.double(), w4.double(), padding=k // 2).permute(0, 2, 3, 1).reshape(-1, cout).cuda()
    x = x4.permute(0, 2, 3, 1).reshape(-1, cin).contiguous().cuda()
    w = w4.permute(0, 2, 3, 1).reshape(-1).contiguous().cuda()          # OHWI
    y = torch.empty(lv.rows, cout, device="cuda")
    flops = 2.0 * lv.rows * cin * cout * k * k
    for name, tile in (("fp32 64x64", 0x203), ("x3   64x64", 0x203 | K.X3), ("fp32 128x64", 0x202), ("x3   128x64", 0x202 | K.X3),
                       ("fp32 128x128", 0x201), ("x3   128x128", 0x201 | K.X3)):
        K.conv_fwd(g, x, w, None, y, tile=tile | (1 << 12))
        err = (y.double() - ref).abs().max().item() / ref.abs().max().item()
        rms = ((y.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
        best = 1e9
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _b in range(5):
                K.conv_fwd(g, x, w, None, y, tile=tile | (1 << 12))
            e.record(); e.synchronize()
            best = min(best, s.elapsed_time(e) / 5)
        print(f"{cin:5d}->{cout:4d} k{k} M={lv.rows:6d} {name:13s}: {best * 1e3:7.1f} us {flops / best / 1e9:7.1f} TFLOP/s   max err {err:.2e}  rms err {rms:.2e}")

run(256, 256, 3, 80, 80)
run(1024, 256, 1, 30, 40)
run(256, 256, 3, 30, 40)
run(128, 512, 1, 60, 80)


def run_wgrad(cin, cout, k, H, W, B=4):
    lv = K.Levels([(H, W)], B)
    gen = torch.Generator().manual_seed(1)
    x4 = torch.randn(B, cin, H, W, generator=gen)
    dy4 = torch.randn(B, cout, H, W, generator=gen)
    ref = torch.nn.grad.conv2d_weight(x4.double(), (cout, cin, k, k), dy4.double(), padding=k // 2).permute(0, 2, 3, 1).reshape(cout, -1).cuda()
    x = x4.permute(0, 2, 3, 1).reshape(-1, cin).contiguous().cuda()
    dy = dy4.permute(0, 2, 3, 1).reshape(-1, cout).contiguous().cuda()
    flops = 2.0 * lv.rows * cin * cout * k * k
    for name, x3 in (("fp32", False), ("x3  ", True)):
        g = K.ConvGeom(lv, cin, cout, k, 1, k // 2)
        g.x3 = x3
        slabs = torch.empty(g.nsplit * cout * k * k * cin, device="cuda")
        K.conv_wgrad(g, dy, x, slabs)
        gw = slabs.view(g.nsplit, cout, k * k * cin).double().sum(0)
        err = (gw - ref).abs().max().item() / ref.abs().max().item()
        rms = ((gw - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
        best = 1e9
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _b in range(5):
                K.conv_wgrad(g, dy, x, slabs)
            e.record(); e.synchronize()
            best = min(best, s.elapsed_time(e) / 5)
        print(f"wgrad {cin:5d}->{cout:4d} k{k} M={lv.rows:6d} S={g.nsplit:2d} {name}: {best * 1e3:7.1f} us {flops / best / 1e9:7.1f} TFLOP/s   max err {err:.2e}  rms err {rms:.2e}")


run_wgrad(256, 256, 3, 80, 80)
run_wgrad(1024, 256, 1, 30, 40)
run_wgrad(256, 256, 3, 30, 40)
run_wgrad(128, 512, 1, 60, 80)
