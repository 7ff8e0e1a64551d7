"""Time radet_nms (vote mode) on synthetic candidates: python tools/bench_nms.py [N per image] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from radet_amd import kernels as K
from radet_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4420
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
g = torch.Generator().manual_seed(0)
c = torch.rand(B, N, 2, generator=g) * torch.tensor([600.0, 440.0])
wh = torch.rand(B, N, 2, generator=g) * 120 + 8
boxes = torch.cat([c - wh / 2, c + wh / 2], -1).cuda().contiguous()
sc = torch.rand(B, N, generator=g).cuda()
lab = torch.randint(0, 21, (B, N), generator=g).cuda()
cnt = torch.full((B,), N, dtype=torch.int32).cuda()
cap = N
ob = torch.empty(B, 100, 4).cuda(); osc = torch.empty(B, 100).cuda()
ol = torch.empty(B, 100, dtype=torch.long).cuda(); oc = torch.zeros(B, dtype=torch.int32).cuda()
aux = torch.empty(2, B * cap, dtype=torch.long).cuda()
ws = torch.empty(K.nms_ws_bytes(B, cap), dtype=torch.uint8).cuda()


def run():
    K.nms(boxes, sc, sc, lab, cnt, B, cap, 0, 0.65, False, 0.025, 100, ob, osc, ol, oc, aux[0], aux[1], ws)


for _ in range(3):
    run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    run()
e.record(); e.synchronize()
print(f"N={N} B={B} {s.elapsed_time(e) / 10 * 1e3:.0f} us per launch, heads {oc.tolist()[:4]}")
