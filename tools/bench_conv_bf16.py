"""TFLOP/s of the implicit-GEMM kernel in its three arithmetic modes on the head-tower shape:
fp32 (v_mfma_f32_32x32x2_f32), bf16 math on fp32 tensors (0x400), bf16 storage (0x800, v_mfma_f32_32x32x16_bf16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radet_amd import kernels as K

def run(B, hw, cin, cout, k, mode, tiles):
    lv = K.Levels(hw, B)
    g = K.ConvGeom(lv, cin, cout, k, 1, k // 2)
    M = lv.rows
    dt = torch.bfloat16 if mode == 0x800 else torch.float32
    x = torch.randn(M, cin, device="cuda").to(dt)
    w = (torch.randn(cout, k * k, cin, device="cuda") * 0.02).to(dt)
    y = torch.empty(M, cout, device="cuda", dtype=dt)
    fl = 2.0 * M * cin * cout * k * k
    out = []
    for t in tiles:
        f = lambda: K.conv_fwd(g, x, w, None, y, relu=True, tile=t | mode)
        for _ in range(3): f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): f()
        e.record(); e.synchronize()
        out.append(f"t{t:#x}:{fl / (s.elapsed_time(e) / 10 * 1e-3) / 1e12:7.1f}")
    return " ".join(out)

TILES = [1, 2, 3, 0x201, 0x202, 0x203]
HW5 = [(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)]
for name, B, hw, cin, cout, k in (("tower 5lvl B4", 4, HW5, 256, 256, 3), ("tower big M=131072", 1, [(512, 256)], 256, 256, 3),
                                  ("l3 1x1 1024->256 M=4800", 4, [(30, 40)], 1024, 256, 1), ("l2 3x3 128 M=19200", 4, [(60, 80)], 128, 128, 3)):
    for mode, mn in ((0, "fp32        "), (0x400, "bf16 math   "), (0x800, "bf16 storage")):
        print(f"{name:26s} {mn} {run(B, hw, cin, cout, k, mode, TILES)}")


def run_wgrad(B, hw, cin, cout, k, mode):
    from radet_amd import _lib
    lv = K.Levels(hw, B)
    g = K.ConvGeom(lv, cin, cout, k, 1, k // 2)
    M = lv.rows
    dt = torch.bfloat16 if mode == 2 else torch.float32
    x = torch.randn(M, cin, device="cuda").to(dt)
    dy = torch.randn(M, cout, device="cuda").to(dt)
    fl = 2.0 * M * cin * cout * k * k
    out = []
    for tflag, tname, t in ((2 << 4, "64x64", 64), (1 << 4, "128x128", 128)):
        tiles = -(-cout // t) * -(-cin // t) * k * k
        for blocks in (512, 1024):
            S = max(1, min(64, round(blocks / tiles)))
            slabs = torch.empty(S * cout * k * k * cin, device="cuda")
            f = lambda: _lib.call("radet_conv2d_wgrad", K._ptr(dy), K._ptr(x), K._ptr(slabs), None, K._ptr(g.fwd_table), M, cin,
                                  cout, cout, k, k, S, mode | tflag | 0x40, K._stream())
            for _ in range(3): f()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): f()
            e.record(); e.synchronize()
            out.append(f"{tname} S={S}:{fl / (s.elapsed_time(e) / 10 * 1e-3) / 1e12:7.1f}")
    return "  ".join(out)


print()
for name, B, hw, cin, cout, k in (("tower 5lvl B4", 4, HW5, 256, 256, 3), ("l3 1x1 1024->256 M=4800", 4, [(30, 40)], 1024, 256, 1),
                                  ("l2 3x3 128 M=19200", 4, [(60, 80)], 128, 128, 3)):
    for mode, mn in ((0, "fp32        "), (1, "bf16 math   "), (2, "bf16 storage")):
        print(f"wgrad {name:24s} {mn} {run_wgrad(B, hw, cin, cout, k, mode)}")
