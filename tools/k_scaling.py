"""Fixed cost of an implicit-GEMM launch: M = B*H*W rows, N = cout, K swept; 20 back-to-back launches per timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radet_amd import kernels as K
H, W, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tile = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0x203
lv = K.Levels([(H, W)], 4)
KS = [int(v) for v in os.environ.get('KS', '32,64,128,256,512,1024,2048').split(',')]
for cin in KS:
    g = K.ConvGeom(lv, cin, cout, 1, 1, 0)
    x = torch.randn(lv.rows, cin, device="cuda")
    w = torch.randn(cout * cin, device="cuda") * 0.05
    y = torch.empty(lv.rows, cout, device="cuda")
    for _ in range(5):
        K.conv_fwd(g, x, w, None, y, relu=True, tile=tile)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            K.conv_fwd(g, x, w, None, y, relu=True, tile=tile)
        e.record(); e.synchronize()
        best = min(best, s.elapsed_time(e) / 20)
    fl = 2.0 * lv.rows * cin * cout
    print(f"M={lv.rows} N={cout} K={cin:5d}: {best * 1e3:6.1f} us  {fl / best / 1e9:6.1f} TF")
