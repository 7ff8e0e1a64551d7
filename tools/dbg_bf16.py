"""Per-layer gradient error of the bf16 math mode vs the bf16 oracle (and vs fp32) -- debugging aid."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_configs as t
from oracle import model as om, synth
from radet_amd.apis import wrap_fp16_model

H, W = 224, 224
img, gt_b, gt_l, p2g, pw = t.batch(H, W, 2)
res = {}
for mode in ("fp32", "bf16"):
    det = t.make(50)
    if mode == "bf16":
        wrap_fp16_model(det)
    det.train()
    losses = det(img=img.cuda(), img_metas=synth.img_metas(2, H, W), return_loss=True, gt_bboxes=gt_b, gt_labels=gt_l,
                 points_to_gt_index=p2g, points_weight=pw)
    sum(losses.values()).backward()
    res[mode] = ({k: v.item() for k, v in losses.items()},
                 {n: p.grad.detach().cpu().double() for n, p in det.named_parameters() if p.requires_grad})
ores = {}
for mode in ("fp32", "bf16"):
    odet = om.OracleDetector(50, seed=1, math=mode)
    ol = odet.forward_train(img, gt_b, gt_l, p2g, pw)
    om.parse_losses(ol).backward()
    ores[mode] = ({k: v.item() for k, v in ol.items()}, {n: g.double() for n, g in odet.named_grads().items()})
print("losses gpu", res, file=sys.stderr) if False else None
for m in ("fp32", "bf16"):
    print(m, "gpu", res[m][0], "oracle", ores[m][0])
names = [n for n in res["bf16"][1] if n.endswith("weight")]
print(f"{'layer':50s} {'gpuB-orB':>9s} {'gpuF-orF':>9s} {'orB-orF':>9s} {'gpuB-gpuF':>9s}")
for n in names[::6] + [x for x in names if x.startswith("neck") or "bbox_head" in x]:
    d = lambda a, b: (a - b).norm().item() / max(b.norm().item(), 1e-12)
    print(f"{n:50s} {d(res['bf16'][1][n], ores['bf16'][1][n]):9.2e} {d(res['fp32'][1][n], ores['fp32'][1][n]):9.2e} "
          f"{d(ores['bf16'][1][n], ores['fp32'][1][n]):9.2e} {d(res['bf16'][1][n], res['fp32'][1][n]):9.2e}")

# forward features
feats = {}
for mode in ("fp32", "bf16"):
    det = t.make(50)
    if mode == "bf16":
        wrap_fp16_model(det)
    det.eval()
    with torch.no_grad():
        feats["gpu", mode] = [f.cpu().double() for f in det.extract_feat(img.cuda())]
        odet = om.OracleDetector(50, seed=1, math=mode)
        with om.conv_math(mode):
            feats["or", mode] = [f.double() for f in odet.extract_feat(img)]
d = lambda a, b: (a - b).norm().item() / max(b.norm().item(), 1e-12)
for l in range(5):
    print(f"P{l+3}: gpuB-orB {d(feats['gpu','bf16'][l], feats['or','bf16'][l]):.2e}  gpuF-orF {d(feats['gpu','fp32'][l], feats['or','fp32'][l]):.2e}"
          f"  orB-orF {d(feats['or','bf16'][l], feats['or','fp32'][l]):.2e}  gpuB-gpuF {d(feats['gpu','bf16'][l], feats['gpu','fp32'][l]):.2e}")
