"""Time radet_decode_candidates on synthetic head outputs: python tools/bench_decode.py [cls_mean] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radet_amd import kernels as K
cls_mean = float(sys.argv[1]) if len(sys.argv) > 1 else -4.0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
HW = [(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)]
lv = K.Levels(HW, B)
ld, nl = K.level_desc(lv, (8, 16, 32, 64, 128))
g = torch.Generator().manual_seed(0)
R = lv.rows
fc = (torch.randn(R, 21, generator=g) * 1.5 + cls_mean).cuda()
fr = torch.relu(torch.randn(R, 4, generator=g) * 2 + 2.5).cuda()
fi = torch.randn(R, generator=g).cuda()
cap = 5000
boxes, scores = torch.zeros(B, cap, 4).cuda(), torch.zeros(B, cap).cuda()
ctr, labels = torch.zeros(B, cap).cuda(), torch.zeros(B, cap, dtype=torch.long).cuda()
count = torch.zeros(B, dtype=torch.int32).cuda()
ws = torch.empty(K.decode_ws_bytes(B, 5, 1000), dtype=torch.uint8).cuda()
hw = torch.tensor([[480., 640.]] * B).cuda()
one = torch.ones(5).cuda()
def run():
    K.decode_candidates(fc, fr, fi, one, ld, nl, B, 21, 0.05, 1000, hw, None, boxes, scores, ctr, labels, count, ws)
for _ in range(3): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): run()
e.record(); e.synchronize()
print(f"cls_mean={cls_mean} B={B} slow={os.environ.get('RADET_DECODE_SLOW', '0')}: {s.elapsed_time(e) / 20 * 1e3:.0f} us, candidates {count.tolist()[:3]}")

