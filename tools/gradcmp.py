import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from tests.test_gpu_configs import make, batch
from tests._grads import grad_rel_errors
from oracle import model as om, synth
for math in ("fp32", "fp32-mfma"):
    os.environ["RADET_MATH"] = math
    for depth, H, W in [(101, 200, 264), (50, 224, 224)]:
        det = make(depth)
        img, gt_b, gt_l, p2g, pw = batch(H, W, 2)
        det.train()
        losses = det(img=img.cuda(), img_metas=synth.img_metas(2, H, W), return_loss=True, gt_bboxes=gt_b, gt_labels=gt_l, points_to_gt_index=p2g, points_weight=pw)
        sum(losses.values()).backward()
        odet = om.OracleDetector(depth, seed=1)
        ol = odet.forward_train(img, gt_b, gt_l, p2g, pw)
        om.parse_losses(ol).backward()
        errs, tot = grad_rel_errors({n: p.grad for n, p in det.named_parameters() if p.requires_grad}, odet.named_grads())
        r = sorted(((d / max(b, 1e-30), n, d, b) for n, (d, b) in errs.items()), reverse=True)
        print(math, depth, H, W, "total", tot, "median rel", np.median([x[0] for x in r]))
        for x in r[:6]: print("   %.2e %s d=%.3e n=%.3e" % x)
