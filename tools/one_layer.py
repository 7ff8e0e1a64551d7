"""Run one conv layer shape repeatedly (for rocprofv3 --pmc): one_layer.py cin cout k H W [res] [tile] [mode=fwd|wgrad]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radet_amd import kernels as K
cin, cout, k, H, W = (int(v) for v in sys.argv[1:6])
res = len(sys.argv) > 6 and sys.argv[6] == "1"
tile = int(sys.argv[7], 0) if len(sys.argv) > 7 else 0
mode = sys.argv[8] if len(sys.argv) > 8 else "fwd"
lv = K.Levels([(H, W)], 4)
g = K.ConvGeom(lv, cin, cout, k, 1, k // 2)
x = torch.randn(lv.rows, cin, device="cuda")
w = torch.randn(cout * k * k * cin, device="cuda") * 0.05
y = torch.empty(lv.rows, cout, device="cuda")
add = torch.randn(lv.rows, cout, device="cuda") if res else None
if mode == "wgrad":
    K.load_tune_cache(); K.autotune_wgrad(g)
    slabs = torch.empty(g.nsplit * cout * k * k * cin, device="cuda")
    fn = lambda: K.conv_wgrad(g, y, x, slabs, None)
    y.normal_()
else:
    if not tile:
        K.load_tune_cache(); K.autotune(g, need_dgrad=False); tile = g.fwd_tile
    fn = lambda: K.conv_fwd(g, x, w, None, y, addend=add, relu=True, tile=tile)
for _ in range(20):
    fn()
torch.cuda.synchronize()
print("tile", hex(tile), "nsplit", g.nsplit, "flags", hex(g.wgrad_flags))
