"""TFLOP/s of one conv shape as the pixel count M grows (1 round of tiles -> many): separates the fixed cost of a launch
(ramp-up, tail, phase lock-step) from the per-tile efficiency.  python tools/bench_m_sweep.py cin cout k [relu+addend]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radet_amd import kernels as K

cin, cout, k = (int(v) for v in sys.argv[1:4])
res = len(sys.argv) > 4
os.environ["RADET_TUNE_FILE"] = "/tmp/none.json"
for hw in ((30, 40), (60, 40), (60, 80), (120, 80), (120, 160), (240, 160)):
    lv = K.Levels([hw], 4)
    g = K.ConvGeom(lv, cin, cout, k, 1, k // 2)
    K.autotune(g, need_dgrad=False)
    x = torch.randn(lv.rows, cin, device="cuda")
    w = torch.randn(cout * k * k * cin, device="cuda") * 0.05
    y = torch.empty(lv.rows, cout, device="cuda")
    add = torch.randn(lv.rows, cout, device="cuda") if res else None
    fn = lambda: K.conv_fwd(g, x, w, None, y, addend=add, relu=True)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30):
        fn()
    e.record(); e.synchronize()
    us = s.elapsed_time(e) / 30 * 1e3
    fl = 2.0 * lv.rows * cin * cout * k * k
    print(f"{cin}->{cout} k{k} M={lv.rows:6d}: {us:7.1f} us {fl / us / 1e6:6.1f} TF  tile={g.fwd_tile:#x}  "
          f"bytes/us={(lv.rows * (cin + cout * (2 if res else 1)) * 4) / us / 1e3:.0f} GB/s")
