"""Other BASELINE.json configurations on the GPU: ResNet-101 backbone / non-640x480 inputs (config 5 geometry at a
size the CPU oracle finishes in seconds), empty-gt batches, and the batched-NMS inference branch."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make(depth, test_nms=None):
    from oracle import synth
    from radet_amd.models import build_detector
    from radet_amd.utils import Config
    cfg = Config.fromfile(os.path.join(REPO, "configs", "bop", "r50_ycbv_pbr.py"))
    cfg.model["pretrained"] = None
    cfg.model["backbone"]["depth"] = depth
    if test_nms:
        cfg.test_cfg["nms"] = test_nms
    d = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    synth.fill_state_dict(d.state_dict(), seed=1)
    return d.cuda()


def batch(H, W, B, G=(3, 0)):
    from oracle import assigner as oa, synth
    img = synth.synth_images(5, B, H, W)
    gt_b, gt_l, p2g, pw = [], [], [], []
    for i in range(B):
        g = G[i % len(G)]
        rng = np.random.RandomState(40 + i)
        boxes = np.zeros((g, 4), np.float32)
        masks = np.zeros((g, H, W), np.uint8)
        for k in range(g):
            w, h = rng.randint(30, W // 2), rng.randint(30, H // 2)
            x, y = rng.randint(0, W - w), rng.randint(0, H - h)
            boxes[k] = (x, y, x + w, y + h)
            masks[k, y:y + h, x + w // 4:x + w] = 1
        labels = rng.randint(0, 21, g).astype(np.int64)
        a, wt = oa.assign_points(boxes, labels, masks, (H, W, 3), rng=np.random.RandomState(i))
        gt_b.append(torch.from_numpy(boxes)); gt_l.append(torch.from_numpy(labels))
        p2g.append(torch.from_numpy(a)); pw.append(torch.from_numpy(wt))
    return img, gt_b, gt_l, p2g, pw


@pytest.mark.parametrize("depth,H,W", [(101, 200, 264), (50, 224, 224)])
def test_train_step_vs_oracle(depth, H, W):
    """Odd feature-map sizes (25x33 ... 2x3), an image without gts in the batch, R101."""
    from oracle import model as om, synth
    det = make(depth)
    img, gt_b, gt_l, p2g, pw = batch(H, W, 2)
    det.train()
    losses = det(img=img.cuda(), img_metas=synth.img_metas(2, H, W), return_loss=True, gt_bboxes=gt_b, gt_labels=gt_l,
                 points_to_gt_index=p2g, points_weight=pw)
    sum(losses.values()).backward()
    odet = om.OracleDetector(depth, seed=1)
    ol = odet.forward_train(img, gt_b, gt_l, p2g, pw)
    om.parse_losses(ol).backward()
    for k in ("loss_cls", "loss_bbox", "loss_iou"):
        assert abs(losses[k].item() - ol[k].item()) <= 1e-4 * max(1.0, abs(ol[k].item())), k
    og = odet.named_grads()
    tot = float(np.sqrt(sum(g.double().norm().item() ** 2 for g in og.values())))
    for n, p in det.named_parameters():
        if p.requires_grad:
            a, b = p.grad.double().norm().item(), og[n].double().norm().item()
            assert abs(a - b) <= 1e-3 * b + 1e-6 * tot, (n, a, b)
    # element-wise: every parameter's gradient TENSOR against the oracle's (a norm survives a tap transposition)
    # The median over the parameters is held to 1e-3 here (5e-4 elsewhere): on these tiny maps the engine and the fp32 oracle
    # decide 5 of ~10^7 ReLU masks differently (three of them in layer3 of the R101: every gradient upstream of a flipped mask
    # moves by ~5e-4), and WHICH knife edges flip depends on the tiles the tuner picks on the spot for this geometry -- the
    # median was 5.3e-4 with the round-6 kernels' picks.  The arithmetic itself is held to the fp64 oracle with the engine's
    # masks handed in: tests/test_gpu_model.py::test_gradients_vs_fp64_oracle[r101_200x264_bs2] (median 5.8e-7, torch-fp32's
    # own distance to fp64 on this batch: 1.0e-4).
    from _grads import assert_grads_close
    assert_grads_close({n: p.grad for n, p in det.named_parameters() if p.requires_grad}, og, median_rtol=1e-3)


def _headline_batch(B=4):
    """bench.make_batch's inputs (the GPU assigner is the product path; the oracle gets the same targets)"""
    import bench
    img, boxes, labels, p2g, pw = bench.make_batch(0, B, torch.device("cuda"))
    return (img, [torch.from_numpy(b) for b in boxes], [torch.from_numpy(l) for l in labels],
            [t.cpu() for t in p2g], [t.cpu() for t in pw])


@pytest.mark.parametrize("math", ["fp32", "fp32-mfma"])
def test_headline_size_bs4_640x480_vs_oracle(math):
    """BASELINE configs[1] at its full size -- r50, 640 x 480, bs 4, the bench's own batch -- against the CPU oracle: loss
    triple within 1e-4 and every parameter's gradient TENSOR against the oracle's (tests/_grads.py: 3e-3 per parameter, median
    5e-4 -- a wiring error is of order 1); both fp32
    arithmetics (products from fp16 hi / lo pairs = the default, native fp32 MFMA)."""
    from oracle import model as om, synth
    from _grads import assert_grads_close
    from radet_amd.models import build_detector
    from radet_amd.utils import Config
    cfg = Config.fromfile(os.path.join(REPO, "configs", "bop", "r50_ycbv_pbr.py"))
    cfg.model["pretrained"] = None
    det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    synth.fill_state_dict(det.state_dict(), seed=0)
    det = det.cuda().train()
    rt = det.runtime(math=math)
    img, gt_b, gt_l, p2g, pw = _headline_batch()
    tg = rt.pack_targets(gt_b, gt_l, [t.cuda() for t in p2g], [t.cuda() for t in pw])
    rt.forward(img)
    losses = rt.loss(tg).clone().cpu()
    rt.backward()
    torch.cuda.synchronize()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    odet = om.OracleDetector(50, seed=0)
    ol = odet.forward_train(img.cpu(), gt_b, gt_l, p2g, pw)
    om.parse_losses(ol).backward()
    for k, a in zip(("loss_cls", "loss_bbox", "loss_iou"), losses.tolist()):
        assert abs(a - ol[k].item()) <= 1e-4 * max(1.0, abs(ol[k].item())), (k, a, ol[k].item())
    mine = {n: rt.flat.g[n] for n in rt.flat.train_names}             # views of the gradient arena (after un-folding)
    worst = assert_grads_close(mine, odet.named_grads())
    print("worst parameter (||d||, ||g||):", worst)


@pytest.mark.timeout(1500)
def test_r101_800x800_bs2_losses_and_gradients_vs_oracle():
    """BASELINE configs[4] at its full size (R101, 800 x 800, bs 2 -- 100 x 100 / 50 x 50 / 25 x 25 / 13 x 13 / 7 x 7 levels,
    their own tiles, split-K choices and gather tables): loss triple AND every parameter's gradient tensor against the CPU
    oracle's forward + backward (tests/_grads.py: per parameter ||g - g_ref|| <= 3e-3 ||g_ref||, median <= 5e-4)."""
    from oracle import model as om, synth
    from _grads import assert_grads_close
    det = make(101)
    img, gt_b, gt_l, p2g, pw = batch(800, 800, 2, G=(4, 2))
    det.train()
    losses = det(img=img.cuda(), img_metas=synth.img_metas(2, 800, 800), return_loss=True, gt_bboxes=gt_b, gt_labels=gt_l,
                 points_to_gt_index=p2g, points_weight=pw)
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    odet = om.OracleDetector(101, seed=1)
    ol = odet.forward_train(img, gt_b, gt_l, p2g, pw)
    om.parse_losses(ol).backward()
    for k in ("loss_cls", "loss_bbox", "loss_iou"):
        assert abs(losses[k].item() - ol[k].item()) <= 1e-4 * max(1.0, abs(ol[k].item())), (k, losses[k].item(), ol[k].item())
    mine = {n: p.grad for n, p in det.named_parameters() if p.requires_grad}
    og = odet.named_grads()
    assert set(mine) == set(og)                                   # every trainable R101 parameter has a partner
    worst = assert_grads_close(mine, og)
    print("R101 800x800 worst parameter (||d||, ||g||):", worst, "of", len(mine))


def test_empty_batch_and_batched_nms_branch():
    from oracle import model as om, synth
    det = make(50, test_nms=dict(type="nms", iou_threshold=0.5))
    H, W = 160, 192
    img, gt_b, gt_l, p2g, pw = batch(H, W, 2, G=(0, 0))
    det.train()
    losses = det(img=img.cuda(), img_metas=synth.img_metas(2, H, W), return_loss=True, gt_bboxes=gt_b, gt_labels=gt_l,
                 points_to_gt_index=p2g, points_weight=pw)
    assert losses["loss_bbox"].item() == 0.0 and losses["loss_iou"].item() == 0.0 and losses["loss_cls"].item() > 0
    sum(losses.values()).backward()
    assert float(dict(det.named_parameters())["bbox_head.atss_reg.weight"].grad.abs().sum()) == 0.0
    det.eval()
    res = det(img=[img.cuda()], img_metas=[synth.img_metas(2, H, W)], return_loss=False, rescale=True)
    odet = om.OracleDetector(50, seed=1, test_cfg=dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, max_per_img=100,
                                                       nms=dict(type="nms", iou_threshold=0.5)))
    ref = odet.simple_test(img, synth.img_metas(2, H, W))
    for per_cls, (rb, rl) in zip(res, ref):
        dets = np.concatenate([np.concatenate([d, np.full((d.shape[0], 1), c, np.float32)], 1) for c, d in enumerate(per_cls)], 0)
        assert dets.shape[0] == rb.shape[0]
        # hard NMS on ~5000 network-produced candidates is knife-edge sensitive (an IoU within 1e-6 of the threshold
        # flips a keep decision), so the end-to-end check matches detections as a set; the NMS kernel itself is
        # checked bit-exactly on fixed inputs in test_gpu_kernels.py::test_nms_ops_bit_exact
        matched = 0
        for row, lab in zip(rb, rl):
            d = np.abs(dets[:, :5] - row[None, :]).max(1)
            j = int(d.argmin())
            matched += int(d[j] <= 2e-3 + 1e-4 * np.abs(row).max() and int(dets[j, 5]) == int(lab))
        assert matched >= 0.95 * rb.shape[0], (matched, rb.shape[0])


def test_bf16_config3_vs_oracle(monkeypatch):
    """BASELINE config 3 arithmetic (mixed precision): conv operands rounded to bf16 into the matrix cores, fp32
    accumulate, fp32 GroupNorm / loss.  Oracle = the same rounding on the CPU (oracle.model.conv_math).

    Rounding is discontinuous: an fp32-level difference (1e-6) between two correct implementations flips a bf16
    rounding now and then, and every flip injects a full bf16 step, so the two sides decorrelate with depth
    (measured: 2.6e-4 after layer1, 3e-3 = the size of the bf16 perturbation itself after layer3).  The exact check of
    the arithmetic is therefore at kernel level (tests/test_gpu_kernels.py, bf16math cases: 1e-5 against the
    convolution of pre-rounded tensors); here we check (a) the first stage agrees far below the bf16 perturbation,
    (b) losses within 3e-3 (the size of one bf16 perturbation), (c) every gradient tensor is within the bf16-vs-fp32 perturbation of the oracle's, and
    closer in aggregate, (d) the mode is really on (results differ from fp32)."""
    from oracle import model as om, synth
    from radet_amd.apis import wrap_fp16_model
    # this geometry is not in the shipped tune file: with the start-up tuner on, the tile / split-K choice (= the fp32
    # summation order, = which bf16 roundings flip) would depend on this run's timings and the loss would move by ~1e-3
    # from run to run; launcher heuristics make the comparison reproducible
    monkeypatch.setenv("RADET_AUTOTUNE", "0")
    H, W = 224, 224
    img, gt_b, gt_l, p2g, pw = batch(H, W, 2)
    res, c2 = {}, {}
    for mode in ("fp32", "bf16"):
        det = make(50)
        if mode == "bf16":
            wrap_fp16_model(det, mode="bf16")
        det.train()
        losses = det(img=img.cuda(), img_metas=synth.img_metas(2, H, W), return_loss=True, gt_bboxes=gt_b,
                     gt_labels=gt_l, points_to_gt_index=p2g, points_weight=pw)
        assert det.runtime().engine.math == (1 if mode == "bf16" else 0)
        sum(losses.values()).backward()
        res[mode] = ({k: v.item() for k, v in losses.items()},
                     {n: p.grad.detach().cpu().double() for n, p in det.named_parameters() if p.requires_grad})
        with torch.no_grad():
            c2[mode] = det.backbone(img.cuda())[0].cpu().double()
    ores = {}
    for mode in ("fp32", "bf16"):
        odet = om.OracleDetector(50, seed=1, math=mode)
        ol = odet.forward_train(img, gt_b, gt_l, p2g, pw)
        om.parse_losses(ol).backward()
        with torch.no_grad(), om.conv_math(mode):
            oc2 = om.backbone(odet.sd, img, 50)[0].double()
        ores[mode] = ({k: v.item() for k, v in ol.items()}, {n: g.double() for n, g in odet.named_grads().items()}, oc2)
    rel = lambda a, b: (a - b).norm().item() / max(b.norm().item(), 1e-30)  # noqa: E731
    # (a) first stage (10 bf16 convs deep)
    gap_c2 = rel(ores["bf16"][2], ores["fp32"][2])
    assert gap_c2 > 1e-3 and rel(c2["bf16"], ores["bf16"][2]) < 0.25 * gap_c2, (rel(c2["bf16"], ores["bf16"][2]), gap_c2)
    # (b) losses
    lb, gb = res["bf16"]
    for k in ("loss_cls", "loss_bbox", "loss_iou"):
        assert abs(lb[k] - ores["bf16"][0][k]) <= 3e-3 * max(1.0, abs(ores["bf16"][0][k])), (k, lb[k], ores["bf16"][0][k])
    # (c) gradients
    ob, of = ores["bf16"][1], ores["fp32"][1]
    tot = float(np.sqrt(sum(g.norm().item() ** 2 for g in ob.values())))
    e2 = g2 = 0.0
    for n, g in gb.items():
        err, gap = (g - ob[n]).norm().item(), (ob[n] - of[n]).norm().item()
        assert err <= 1.5 * gap + 1e-3 * tot, (n, err, gap)
        e2, g2 = e2 + err ** 2, g2 + gap ** 2
    assert e2 < 0.75 ** 2 * g2, (e2 ** 0.5, g2 ** 0.5)
    # (d) the mode is really on
    assert any(abs(lb[k] - res["fp32"][0][k]) > 1e-5 * max(1.0, abs(lb[k])) for k in lb), (lb, res["fp32"][0])
    print(f"bf16 vs oracle-bf16: C2 {rel(c2['bf16'], ores['bf16'][2]):.2e} (bf16 perturbation {gap_c2:.2e}); "
          f"gradient error {e2 ** 0.5:.3f} vs perturbation {g2 ** 0.5:.3f}")


def test_bf16_storage_mode_train_step():
    """math="bf16-storage": bf16 activations / folded weights / activation gradients in HBM.  Same statistical
    criteria as the bf16 math mode (activations are additionally rounded when stored, so the perturbation is larger):
    losses within 1e-2, gradient tensors within the bf16-vs-fp32 perturbation of the oracle's bf16 model, finite, and
    the native step (clip + AdamW) runs."""
    from oracle import model as om, synth
    H, W = 224, 224
    img, gt_b, gt_l, p2g, pw = batch(H, W, 2)
    from radet_amd.apis import wrap_fp16_model
    det = wrap_fp16_model(make(50))             # default mode of the mixed-precision switch
    det.train()
    rt = det.runtime()
    assert rt.engine.h16
    losses = det(img=img.cuda(), img_metas=synth.img_metas(2, H, W), return_loss=True, gt_bboxes=gt_b, gt_labels=gt_l,
                 points_to_gt_index=p2g, points_weight=pw)
    assert det.runtime().engine.h16
    sum(losses.values()).backward()
    grads = {n: p.grad.detach().cpu().double() for n, p in det.named_parameters() if p.requires_grad}
    ores = {}
    for mode in ("fp32", "bf16"):
        odet = om.OracleDetector(50, seed=1, math=mode)
        ol = odet.forward_train(img, gt_b, gt_l, p2g, pw)
        om.parse_losses(ol).backward()
        ores[mode] = ({k: v.item() for k, v in ol.items()}, {n: g.double() for n, g in odet.named_grads().items()})
    for k in ("loss_cls", "loss_bbox", "loss_iou"):
        assert abs(losses[k].item() - ores["bf16"][0][k]) <= 1e-2 * max(1.0, abs(ores["bf16"][0][k])), (k, losses[k].item())
    ob, of = ores["bf16"][1], ores["fp32"][1]
    tot = float(np.sqrt(sum(g.norm().item() ** 2 for g in ob.values())))
    e2 = g2 = 0.0
    for n, g in grads.items():
        assert torch.isfinite(g).all(), n
        err, gap = (g - ob[n]).norm().item(), (ob[n] - of[n]).norm().item()
        assert err <= 3.0 * gap + 2e-3 * tot, (n, err, gap)
        e2, g2 = e2 + err ** 2, g2 + gap ** 2
    assert e2 < 2.0 ** 2 * g2, (e2 ** 0.5, g2 ** 0.5)
    print(f"bf16-storage vs oracle-bf16: gradient error {e2 ** 0.5:.3f} vs bf16-math perturbation {g2 ** 0.5:.3f}")
    # native train step
    rt.init_optimizer()
    tg = rt.pack_targets(gt_b, gt_l, p2g, pw)
    out = rt.train_step(img.cuda(), tg)
    assert torch.isfinite(out).all()


def test_bf16_storage_inference_close_to_fp32(monkeypatch):
    """Inference in bf16-storage mode: the head outputs (what decode + NMS consume) stay within bf16 noise of the fp32
    engine's, and the post-processing runs on them.  (Detections themselves are not compared: with random weights the
    ranking of thousands of near-equal scores is decided by that noise.)"""
    from oracle import synth
    # fixed tiles: run-time tuned tiles change which partial sums are rounded to bf16, and with them the worst element
    # of ~40 000 by a few 1e-3 -- the bound below is on that maximum
    monkeypatch.setenv("RADET_AUTOTUNE", "0")
    H, W = 224, 224
    img = synth.synth_images(5, 2, H, W).cuda()
    metas = synth.img_metas(2, H, W)
    outs = {}
    for mode in ("fp32", "bf16-storage"):
        det = make(50).eval()
        with torch.no_grad():
            det.bbox_head.atss_cls.bias += 2.5
        rt = det.runtime(math=mode)
        dets = rt.detect(img, metas, det.test_cfg, rescale=True)
        b = rt.engine.buf
        outs[mode] = [b[k].float().clone() for k in ("cls", "reg_u", "iou")] + [dets]
        assert all(d.shape[0] > 0 and torch.isfinite(d).all() for d, _ in dets)
    for a_, b_ in zip(outs["fp32"][:3], outs["bf16-storage"][:3]):
        err = (a_ - b_).abs().max().item()
        assert err <= 0.05 * a_.std().item() + 0.03 * a_.abs().max().item(), (err, a_.std().item(), a_.abs().max().item())


def test_head_with_more_than_32_classes():
    """num_classes = 40: the predictor convs do not fit the 32-column LDS-patch kernel and must take the implicit-GEMM
    path (forward, loss, backward, detection) -- against the oracle."""
    from oracle import model as om, synth
    from radet_amd.models import build_detector
    from radet_amd.utils import Config
    cfg = Config.fromfile(os.path.join(REPO, "configs", "bop", "r50_ycbv_pbr.py"))
    cfg.model["pretrained"] = None
    cfg.model["bbox_head"]["num_classes"] = 40
    det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    synth.fill_state_dict(det.state_dict(), seed=2)
    det = det.cuda().train()
    H, W = 160, 192
    img, gt_b, gt_l, p2g, pw = batch(H, W, 2, G=(3, 2))
    gt_l = [l + 17 for l in gt_l]                          # labels up to 37
    losses = det(img=img.cuda(), img_metas=synth.img_metas(2, H, W), return_loss=True, gt_bboxes=gt_b, gt_labels=gt_l,
                 points_to_gt_index=p2g, points_weight=pw)
    sum(losses.values()).backward()
    odet = om.OracleDetector(50, seed=2, num_classes=40)      # same seeded weights by parameter name
    ol = odet.forward_train(img, gt_b, gt_l, p2g, pw)
    om.parse_losses(ol).backward()
    for k in ("loss_cls", "loss_bbox", "loss_iou"):
        assert abs(losses[k].item() - ol[k].item()) <= 1e-4 * max(1.0, abs(ol[k].item())), k
    from _grads import assert_grads_close
    # (K = 48 after padding: the cls predictor's dgrad / wgrad run on the native fp32 MFMA, whose distance to the CPU
    # oracle is ~3e-4 per GEMM -- tools/gradcmp.py -- and every gradient passes through it)
    assert_grads_close({n: p.grad for n, p in det.named_parameters() if p.requires_grad}, odet.named_grads(),
                       rtol=5e-3, median_rtol=2e-3)
    assert det.runtime().engine.pred_cls.cout == 40


@pytest.mark.parametrize("fs", [-1, 0])
def test_trainable_stem_and_layer1_vs_oracle(fs):
    """ResNet(frozen_stages=-1): conv1 / bn1 and layer1 train (resnet.py:572-588) -- dgrad through layer1.0's conv1 and
    projection shortcut, max-pool backward (first maximum of a window in scan order, ties among the ReLU's zeros included),
    the stem's ReLU, the stem weight gradient from the NCHW image; frozen_stages=0: stem frozen, layer1 trainable.  Losses
    and every gradient tensor against the oracle; one optimizer step moves the stem."""
    from oracle import model as om, synth
    from radet_amd.models import build_detector
    from radet_amd.utils import Config
    cfg = Config.fromfile(os.path.join(REPO, "configs", "bop", "r50_ycbv_pbr.py"))
    cfg.model["pretrained"] = None
    cfg.model["backbone"]["frozen_stages"] = fs
    det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    synth.fill_state_dict(det.state_dict(), seed=5)
    det = det.cuda().train()
    H, W = 150, 202                                        # odd sizes: ragged max-pool windows at the right / bottom edge
    img, gt_b, gt_l, p2g, pw = batch(H, W, 2, G=(3, 2))
    losses = det(img=img.cuda(), img_metas=synth.img_metas(2, H, W), return_loss=True, gt_bboxes=gt_b, gt_labels=gt_l,
                 points_to_gt_index=p2g, points_weight=pw)
    sum(losses.values()).backward()
    odet = om.OracleDetector(50, seed=5, frozen_stages=fs)
    ol = odet.forward_train(img, gt_b, gt_l, p2g, pw)
    om.parse_losses(ol).backward()
    for k in ("loss_cls", "loss_bbox", "loss_iou"):
        assert abs(losses[k].item() - ol[k].item()) <= 1e-4 * max(1.0, abs(ol[k].item())), k
    mine = {n: p.grad for n, p in det.named_parameters() if p.requires_grad}
    ref = odet.named_grads()
    assert ("backbone.conv1.weight" in mine) == (fs < 0) and "backbone.layer1.0.conv1.weight" in mine
    assert set(mine) == set(ref)
    from _grads import assert_grads_close
    assert_grads_close(mine, ref)
    if fs < 0:
        rt = det.runtime()
        rt.init_optimizer()
        tg = rt.pack_targets(gt_b, gt_l, p2g, pw)
        w0 = det.backbone.conv1.weight.detach().clone()
        rt.train_step(img.cuda(), tg)
        torch.cuda.synchronize()
        assert not torch.equal(w0, det.backbone.conv1.weight.detach())


def test_trainable_stem_vs_reference_golden():
    """frozen_stages=-1 on the reference's own batch: losses and sampled elements of all 210 gradient tensors (conv1, bn1 and
    layer1 included) as written by the reference (tests/golden/model_grads_stem.npz)."""
    from oracle import synth
    from radet_amd.models import build_detector
    from radet_amd.utils import Config
    g = np.load(os.path.join(REPO, "tests", "golden", "model_grads_stem.npz"))
    a = np.load(os.path.join(REPO, "tests", "golden", "assigner.npz"))
    tags = ("g8", "g3")
    gt_b = [torch.from_numpy(a[t + "_boxes"]) for t in tags]
    gt_l = [torch.from_numpy(a[t + "_labels"]) for t in tags]
    p2g = [torch.from_numpy(a[t + "_p2g"].astype(np.int64)) for t in tags]
    pw = [torch.from_numpy(a[t + "_w"]) for t in tags]
    cfg = Config.fromfile(os.path.join(REPO, "configs", "bop", "r50_ycbv_pbr.py"))
    cfg.model["pretrained"] = None
    cfg.model["backbone"]["frozen_stages"] = -1
    det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    synth.fill_state_dict(det.state_dict(), seed=0)
    det = det.cuda().train()
    img = synth.synth_images(0, 2).cuda()
    losses = det(img=img, img_metas=synth.img_metas(2), return_loss=True, gt_bboxes=gt_b, gt_labels=gt_l,
                 points_to_gt_index=p2g, points_weight=pw)
    for k, ref in zip(("loss_cls", "loss_bbox", "loss_iou"), g["losses"]):
        assert abs(losses[k].item() - float(ref)) <= 1e-4 * max(1.0, abs(float(ref))), k
    sum(losses.values()).backward()
    named = dict(det.named_parameters())
    names = [str(n) for n in g["names"]]
    assert sorted(names) == sorted(n for n, p in named.items() if p.requires_grad)
    from _grads import assert_sampled_grads
    assert_sampled_grads({n: named[n].grad for n in names}, g, atol_total=2e-6, total=float(g["total_grad_norm"]))


def test_stale_data_write_is_refolded():
    """A parameter written through `.data` (its own version counter: the engine cannot see it) is picked up after
    `invalidate_folded_weights()` / a train()-eval() switch; a tracked write (copy_ on the Parameter) by itself."""
    from oracle import synth
    det = make(50).eval()
    img = synth.synth_images(9, 1, 128, 160).cuda()
    with torch.no_grad():
        ref0 = torch.cat([f.reshape(-1) for f in det.extract_feat(img)]).clone()
        w = det.neck.fpn_convs[0].conv.weight
        w.data.mul_(2.0)                                     # untracked
        det.invalidate_folded_weights()
        a = torch.cat([f.reshape(-1) for f in det.extract_feat(img)]).clone()
        assert not torch.equal(a, ref0)
        w.data.mul_(0.5)
        det.train(); det.eval()                              # mode switches fold again
        assert torch.equal(torch.cat([f.reshape(-1) for f in det.extract_feat(img)]), ref0)
        w.mul_(2.0)                                          # tracked by torch: seen without help
        assert torch.equal(torch.cat([f.reshape(-1) for f in det.extract_feat(img)]), a)


def _det_rows(dets, labels):
    d = dets.cpu().numpy() if torch.is_tensor(dets) else np.asarray(dets)
    l = labels.cpu().numpy() if torch.is_tensor(labels) else np.asarray(labels)
    o = np.argsort(-d[:, 4], kind="stable")
    return d[o], l[o]


@pytest.mark.timeout(1500)
def test_streamed_inference_b16_vs_oracle():
    """BASELINE configs[3] the way bench.py / tools/bench_configs.py time it -- batch 16 (the reference config's
    samples_per_gpu), `rt.detect_stream` (head outputs alternating between two buffer sets, decode + NMS of batch k on the
    chain stream next to the forward pass of batch k + 1), the cls bias shifted so that 2 % of the logits pass score_thr
    (~1800 candidates per image into vote-NMS, every image returns max_per_img = 100 boxes) -- against the CPU oracle's
    simple_test (radet_head.py:55-169, vote_ext.cpp:70-207) on three different batches = 48 images.
    A detection agrees when the oracle has one of the same label with box and score within 1e-4 (+ 2e-3 px / 1e-6), which
    also pins the keep order up to swaps of scores closer than that.  Two correct fp32 forward passes differ by ~1e-6 in a
    score, and each image has ~134 000 of them with ~2 700 above score_thr = 0.05: nearly every image holds a score within
    1e-6 of the threshold (the oracle's own margins are measured below), so now and then a candidate is in one candidate set
    and not in the other and moves the vote of its cluster.  Required: every image returns 100 detections of which >= 95 agree,
    and all but at most 3 of the 48 images agree in ALL 100 (measured: 48 of 48, 46 of them position by position -- the other two
    swap neighbours whose scores differ by less than the tolerance)."""
    from oracle import model as om, synth
    from radet_amd.models import build_detector
    from radet_amd.utils import Config
    cfg = Config.fromfile(os.path.join(REPO, "configs", "bop", "r50_ycbv_pbr.py"))
    cfg.model["pretrained"] = None
    torch.manual_seed(0)
    det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().eval()
    rt = det.runtime()
    B, NB = 16, 3
    g = torch.Generator().manual_seed(7)
    batches = [torch.randn(B, 3, 480, 640, generator=g) for _ in range(NB)]
    metas = [dict(img_shape=(480, 640, 3), scale_factor=np.full(4, 1.0 + 0.125 * (i % 3), np.float32)) for i in range(B)]
    rt.detect(batches[0].cuda(), metas, det.test_cfg, rescale=True)
    with torch.no_grad():                      # tools/bench_configs.py::infer's bias shift
        logits = rt.engine.buf["cls"].flatten()
        q = torch.quantile(logits[torch.randperm(logits.numel(), device=logits.device)[:2_000_000]].float(), 0.98)
        det.bbox_head.atss_cls.bias += float(np.log(0.05 / 0.95)) - float(q)
    got = list(rt.detect_stream(((im.cuda(), metas) for im in batches), det.test_cfg, rescale=True))
    assert len(got) == NB and all(len(b) == B for b in got)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    odet = om.OracleDetector(50, seed=None, test_cfg=dict(det.test_cfg))
    sd = det.state_dict()
    with torch.no_grad():
        for n, t in odet.sd.items():
            t.copy_(sd[n].cpu())
    thr = float(det.test_cfg["score_thr"])
    stats = []
    for bi, im in enumerate(batches):
        with torch.no_grad(), om.conv_math(odet.math):
            outs = om.head(odet.sd, odet.extract_feat(im))
        ref = om.get_bboxes(*outs, metas, odet.test_cfg, True)
        for i in range(B):
            margin = min(float((c[i].sigmoid() - thr).abs().min()) for c in outs[0])     # nearest score to score_thr (oracle)
            d, l = _det_rows(*got[bi][i])
            r, rl = _det_rows(*ref[i])
            assert len(r) == 100 and len(d) == 100, (bi, i, len(r), len(d))
            tol = 1e-4 * np.abs(r[:, :5]) + np.array([2e-3] * 4 + [1e-6])
            used, hit = np.zeros(len(d), bool), 0
            for k in range(len(r)):
                cand = np.where(~used & (l == rl[k]) & (np.abs(d[:, :5] - r[k, :5]) <= tol[k]).all(1))[0]
                if len(cand):
                    used[cand[0]] = True
                    hit += 1
            in_order = bool(np.array_equal(l, rl) and (np.abs(d[:, :5] - r[:, :5]) <= tol).all())
            stats.append((hit, in_order, margin))
            assert hit >= 95, (bi, i, hit, margin)
    full = sum(h == 100 for h, _, _ in stats)
    print(f"streamed B=16 inference vs oracle: {len(stats)} images; all 100 detections agree in {full}, position by position in "
          f"{sum(o for _, o, _ in stats)}; fewest agreeing {min(h for h, _, _ in stats)}; oracle's nearest score to score_thr: median "
          f"{np.median([m for _, _, m in stats]):.1e}; images below 100: {[(h, f'{m:.0e}') for h, _, m in stats if h < 100]}")
    assert full >= len(stats) - 3
