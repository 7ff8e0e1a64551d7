"""GroupNorm(32, 256)+ReLU forward / backward over the head's multi-level buffer, alone on the device: microseconds and
effective HBM GB/s (algorithmic bytes: fwd 2 reads + 1 write, bwd 4 reads + 1 write of [R, 256]).
python tools/bench_gn.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radet_amd import kernels as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lv = K.Levels([(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)], B)
R = lv.rows
dev = "cuda"
for dt in (torch.float32, torch.bfloat16):
    z = torch.randn(R, 256, device=dev).to(dt)
    dy = torch.randn(R, 256, device=dev).to(dt)
    y, dz = torch.empty_like(z), torch.empty_like(z)
    gamma, beta = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev) * 0.1
    stats = torch.empty(5 * B * 64, device=dev)
    ws = torch.empty(K.gn_ws_floats(lv), device=dev)
    dg, db = torch.empty(256, device=dev), torch.empty(256, device=dev)

    def timeit(fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        e.synchronize()
        return s.elapsed_time(e) / n * 1e3

    es = z.element_size()
    tf = timeit(lambda: K.gn_relu_fwd(lv, z, gamma, beta, y, stats, ws))
    tb = timeit(lambda: K.gn_relu_bwd(lv, dy, z, stats, gamma, beta, dz, dg, db, ws))
    print(f"{dt}: R={R} fwd {tf:.1f} us = {3 * R * 256 * es / tf / 1e3:.0f} GB/s | bwd {tb:.1f} us = {5 * R * 256 * es / tb / 1e3:.0f} GB/s")
