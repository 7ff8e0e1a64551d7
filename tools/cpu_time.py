"""Host enqueue time of one train step vs its GPU time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from radet_amd.models import build_detector
from radet_amd.utils import Config
cfg = Config.fromfile("configs/bop/r50_ycbv_pbr.py"); cfg.model["pretrained"] = None
det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
rt = det.runtime(); rt.init_optimizer()
img, boxes, labels, p2g, pw = bench.make_batch(0, 4, torch.device("cuda"))
tg = rt.pack_targets([torch.from_numpy(b) for b in boxes], [torch.from_numpy(l) for l in labels], list(p2g), list(pw))
for _ in range(3): rt.train_step(img, tg)
torch.cuda.synchronize()
for streams in (True, False):
    rt.engine.use_streams = streams
    for _ in range(2): rt.train_step(img, tg)
    torch.cuda.synchronize()
    enq, tot = [], []
    for _ in range(5):
        t0 = time.perf_counter(); rt.train_step(img, tg); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        enq.append(t1 - t0); tot.append(t2 - t0)
    print(f"streams={streams}: host enqueue {np.median(enq)*1e3:.2f} ms, step {np.median(tot)*1e3:.2f} ms")
