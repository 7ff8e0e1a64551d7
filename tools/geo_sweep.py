"""Train steps and detect calls over several geometries (tuned and untuned): finite losses, img/s per geometry."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from radet_amd.models import build_detector
from radet_amd.utils import Config
from radet_amd.datasets import LabelAssignment
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = Config.fromfile(os.path.join(ROOT, "configs", "bop", "r50_ycbv_pbr.py")); cfg.model["pretrained"] = None
torch.manual_seed(0)
det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
rt = det.runtime(); rt.init_optimizer(); rt.set_loss_from_head(det.bbox_head)
la = LabelAssignment(neg_threshold=0.2, positive_num=10, balance_sample=True)
for (B, H, W) in [(1, 480, 640), (2, 480, 640), (8, 480, 640), (2, 512, 512), (3, 352, 608), (4, 480, 640)]:
    g = torch.Generator().manual_seed(0)
    img = torch.randn(B, 3, H, W, generator=g).cuda()
    rng = np.random.RandomState(0)
    boxes, labels, masks = [], [], []
    for i in range(B):
        b, l, m = bench.synth_objects(rng, 5, H, W); boxes.append(b); labels.append(l); masks.append(m)
    p2g, pw = la.assign_batch(boxes, masks, (H, W, 3), rngs=[np.random.RandomState(i) for i in range(B)])
    tg = rt.pack_targets([torch.from_numpy(b) for b in boxes], [torch.from_numpy(l) for l in labels], list(p2g), list(pw))
    for _ in range(3): out = rt.train_step(img, tg)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): out = rt.train_step(img, tg)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    assert torch.isfinite(out).all()
    print(f"train B={B} {W}x{H}: {dt*1e3:.2f} ms/step {B/dt:.1f} img/s losses {out.cpu().numpy()}")
det.eval()
for (B, H, W) in [(1, 480, 640), (16, 480, 640), (5, 320, 416)]:
    img = torch.randn(B, 3, H, W).cuda()
    metas = [dict(img_shape=(H, W, 3), scale_factor=np.ones(4, np.float32)) for _ in range(B)]
    for _ in range(2): o = rt.detect(img, metas, det.test_cfg, rescale=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): o = rt.detect(img, metas, det.test_cfg, rescale=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"detect B={B} {W}x{H}: {dt*1e3:.2f} ms {B/dt:.1f} img/s dets {[int(d.shape[0]) for d, _ in o][:4]}")
