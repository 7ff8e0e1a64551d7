"""Size-independent properties at BASELINE's full sizes (r50_ycbv_pbr, 640x480, batch 4), where the CPU oracle is too
slow to be the checker for every case: exact linearity of the tower GEMM, idempotence of hard NMS, conservation laws of
the assigner, sortedness of the NMS output, determinism of the train step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HW5 = [(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)]


def test_tower_gemm_is_exactly_linear_in_powers_of_two():
    """conv(2x) == 2 conv(x) and conv(x, W/4) == conv(x, W)/4 bit for bit (scaling by powers of two commutes with every
    fp32 rounding), on the full B=4 tower shape incl. the tail-split tiles; fwd, dgrad and wgrad."""
    from radet_amd import kernels as K
    lv = K.Levels(HW5, 4)
    g = K.ConvGeom(lv, 256, 256, 3, 1, 1)
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(lv.rows, 256, generator=gen).cuda()
    w = (torch.randn(256, 9, 256, generator=gen) * 0.02).cuda()
    y1, y2 = torch.empty(lv.rows, 256, device="cuda"), torch.empty(lv.rows, 256, device="cuda")
    for tile in (0x203, 0x203 | K.STAGES3, 1):
        K.conv_fwd(g, x, w, None, y1, tile=tile)
        K.conv_fwd(g, 2 * x, w / 4, None, y2, tile=tile)
        assert torch.equal(y2 * 2, y1)
    K.conv_dgrad(g, x, w, y1, tile=0x203)
    K.conv_dgrad(g, x * 0.5, w * 8, y2, tile=0x203)
    assert torch.equal(y2 * 0.25, y1)
    S = g.nsplit
    s1, s2 = torch.empty(S, 256, 9, 256, device="cuda"), torch.empty(S, 256, 9, 256, device="cuda")
    dy = torch.randn(lv.rows, 256, generator=gen).cuda()
    K.conv_wgrad(g, dy, x, s1)
    K.conv_wgrad(g, dy * 4, x * 0.5, s2)
    assert torch.equal(s2 * 0.5, s1)


def test_hard_nms_idempotent_and_sorted():
    from radet_amd import ops
    rs = np.random.RandomState(0)
    n = 5000
    c = rs.rand(n, 2) * np.array([600.0, 440.0])
    wh = rs.rand(n, 2) * 120 + 8
    boxes = torch.from_numpy(np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32))
    scores = torch.from_numpy(rs.rand(n).astype(np.float32))
    labels = torch.from_numpy(rs.randint(0, 21, n).astype(np.int64))
    dets, keep = ops.batched_nms(boxes, scores, labels, dict(type="nms", iou_threshold=0.5))
    assert (dets[:-1, 4] >= dets[1:, 4]).all()                                   # sorted by score
    dets2, keep2 = ops.batched_nms(dets[:, :4], dets[:, 4], labels[keep], dict(type="nms", iou_threshold=0.5))
    assert dets2.shape == dets.shape and torch.equal(dets2, dets)                # NMS(NMS(x)) == NMS(x)
    vb, vl = ops.vote_nms(boxes, scores, labels, dict(type="vote", iou_threshold=0.65, iou_enable=False), score_factor=torch.ones(n),
                          max_num=100)
    assert vb.shape[0] == 100 and (vb[:-1, 4] >= vb[1:, 4]).all()


def test_assigner_conservation_laws_full_size():
    """B=4 at 640x480: every point is negative (-1, weight 1), ignored (0, weight 0) or positive (1..G, integer weight);
    each gt that owns candidates hands out exactly positive_num draws."""
    import bench
    from radet_amd.datasets import LabelAssignment
    rng = np.random.RandomState(5)
    boxes, masks, rngs = [], [], []
    for i in range(4):
        b, _, m = bench.synth_objects(rng, int(rng.randint(1, 9)))
        boxes.append(b); masks.append(m); rngs.append(np.random.RandomState(i))
    la = LabelAssignment(neg_threshold=0.2, positive_num=10, adapt_positive_num=False, balance_sample=True)
    p2g, pw = la.assign_batch(boxes, masks, (480, 640, 3), rngs=rngs)
    p2g, pw = p2g.cpu().numpy(), pw.cpu().numpy()
    assert p2g.shape == (4, 6400)
    for i in range(4):
        G = boxes[i].shape[0]
        a, w = p2g[i], pw[i]
        assert a.min() >= -1 and a.max() <= G
        assert (w[a == -1] == 1).all() and (w[a == 0] == 0).all()
        pos = a > 0
        assert (w[pos] >= 1).all() and (w[pos] == np.round(w[pos])).all()
        per_gt = np.bincount(a[pos], weights=w[pos], minlength=G + 1)[1:]
        assert set(np.unique(per_gt)) <= {0.0, 10.0}, per_gt                     # 10 draws per gt (with multiplicity) or none
        assert per_gt.sum() > 0


def test_train_step_is_deterministic():
    """Two identical runs of the native train step (same tuning) give bit-identical losses and parameters: all
    reductions (split-K, wgrad slabs, GroupNorm, loss, grad norm) use fixed summation orders, no atomics on floats."""
    import os
    import bench
    from radet_amd.models import build_detector
    from radet_amd.utils import Config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for rep in range(2):
        cfg = Config.fromfile(os.path.join(root, "configs", "bop", "r50_ycbv_pbr.py"))
        cfg.model["pretrained"] = None
        torch.manual_seed(0)
        det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
        rt = det.runtime()
        rt.init_optimizer()
        img, boxes, labels, p2g, pw = bench.make_batch(0, 4, torch.device("cuda"))     # the headline batch (bs = 4)
        tg = rt.pack_targets([torch.from_numpy(b) for b in boxes], [torch.from_numpy(l) for l in labels], list(p2g), list(pw))
        for _ in range(2):
            losses = rt.train_step(img, tg).clone()
        outs.append((losses.cpu(), rt.flat.params.clone().cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_frozen_prefix_prefetch_is_bit_identical():
    """train_step(img, tg, next_img=...) computes the frozen stem + layer1 of the next batch during this step's backward pass
    (Engine.prefetch_prefix; frozen_stages of resnet.py:572-588).  Four steps over alternating batches, with a hand-over that
    matches (picked up), one that does not (another tensor than announced: recomputed) and one where the announced tensor was
    overwritten in place in between (version counter: recomputed) -- losses of every step, parameters and AdamW state equal
    the unpipelined run bit for bit."""
    import os
    import bench
    from radet_amd.models import build_detector
    from radet_amd.utils import Config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    batches = []
    for r in range(2):
        img, boxes, labels, p2g, pw = bench.make_batch(r, 2, torch.device("cuda"))
        batches.append((img, boxes, labels, p2g, pw))
    outs = []
    for prefetch in (False, True):
        cfg = Config.fromfile(os.path.join(root, "configs", "bop", "r50_ycbv_pbr.py"))
        cfg.model["pretrained"] = None
        torch.manual_seed(0)
        det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
        rt = det.runtime()
        rt.init_optimizer()
        imgs = [b[0].clone() for b in batches]
        tgs = [rt.pack_targets([torch.from_numpy(x) for x in b[1]], [torch.from_numpy(x) for x in b[2]], list(b[3]), list(b[4]))
               for b in batches]
        losses, picked = [], []
        # step 0 announces batch 1 (picked up), step 1 announces batch 0 but step 2 is given batch 1 (recomputed), step 2 announces
        # batch 0, which is then overwritten in place with its own values (new version: recomputed), step 3 runs on it
        order = [(0, 1), (1, 0), (1, 0), (0, None)]
        for i, (cur, nxt) in enumerate(order):
            if i == 3:
                imgs[0].copy_(batches[0][0])
            before = rt.engine._pfx_ready is not None
            losses.append(rt.train_step(imgs[cur], tgs[cur], next_img=imgs[nxt] if (prefetch and nxt is not None) else None).clone())
            picked.append(before)
        torch.cuda.synchronize()
        if prefetch:
            assert picked == [False, True, True, True]               # a hand-over was pending at the head of steps 1-3 ...
            assert rt.engine._pfx_sets[1] is not None                # ... and the second buffer set exists
        outs.append((torch.stack(losses).cpu(), rt.flat.params.clone().cpu(), rt.opt_state["m"].clone().cpu()))
    for a_, b_ in zip(outs[0], outs[1]):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("math", ["fp32", "bf16-storage"])
def test_tape_replay_is_bit_identical(math):
    """The launch tape (radet_amd/tape.py, include/radet_hip.h "launch tape"): eight train steps over two alternating batches
    (other tensors, other gt counts: the tape's pointer words are patched) with a learning rate that changes every step, once
    eager (RADET_TAPE=0) and once with the tape (three eager steps -- the first one builds the geometry plan --, one recorded, four replayed) -- losses of every step,
    parameters, AdamW moments and the gradient arena are equal bit for bit.  Then what must drop the tape: a write to a frozen
    parameter (the frozen convs are folded outside the tape) and another batch size."""
    import os
    import bench
    from radet_amd.models import build_detector
    from radet_amd.utils import Config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    batches = [bench.make_batch(r, 2, torch.device("cuda")) for r in range(2)]
    outs = []
    for mode in ("0", "1"):
        cfg = Config.fromfile(os.path.join(root, "configs", "bop", "r50_ycbv_pbr.py"))
        cfg.model["pretrained"] = None
        torch.manual_seed(0)
        det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
        rt = det.runtime(math=math)
        rt.tape_mode = mode
        rt.init_optimizer()
        rt.set_loss_from_head(det.bbox_head)
        tgs = [rt.pack_targets([torch.from_numpy(x) for x in b[1]], [torch.from_numpy(x) for x in b[2]], list(b[3]), list(b[4]))
               for b in batches]
        losses = []
        for i in range(8):
            losses.append(rt.train_step(batches[i & 1][0], tgs[i & 1], lr=4e-4 * (1 + 0.1 * i)).clone())
        torch.cuda.synchronize()
        if mode == "1":
            st = rt.tape_stats()
            assert st is not None and st["replays"] == 4 and st["calls"] > 150 and st["segments"] == 1, (st, rt._tape["failed"])
            assert st["bound"]["img"] >= 1 and st["bound"]["p2g"] >= 1 and st["bound"]["boxes"] >= 1, st
        else:
            assert rt.tape_stats() is None
        outs.append((torch.stack(losses).cpu(), rt.flat.params.clone().cpu(), rt.opt_state["m"].clone().cpu(),
                     rt.opt_state["v"].clone().cpu(), rt.flat.grads.clone().cpu()))
    for a_, b_ in zip(outs[0], outs[1]):
        assert torch.equal(a_, b_)
    # (rt is the taped runtime) a frozen parameter moves: the tape is dropped, the step runs eager and re-folds the frozen part
    tape = rt._tape["tape"]
    with torch.no_grad():
        det.backbone.conv1.weight.mul_(1.0)
    rt.train_step(batches[0][0], tgs[0])
    assert rt._tape["tape"] is None and rt._tape["count"] == 1
    for _ in range(4):
        rt.train_step(batches[0][0], tgs[0])
    assert rt._tape["tape"] is not None and rt._tape["tape"] is not tape and rt._tape["tape"].replays == 2
    # another batch size: another plan, its own tape later; the first plan's tape is not replayed on it
    b4 = bench.make_batch(0, 4, torch.device("cuda"))
    tg4 = rt.pack_targets([torch.from_numpy(x) for x in b4[1]], [torch.from_numpy(x) for x in b4[2]], list(b4[3]), list(b4[4]))
    l4 = rt.train_step(b4[0], tg4).clone()
    assert rt._tape["tape"] is None and torch.isfinite(l4).all()


def test_both_fp32_arithmetics_agree_on_losses_and_gradients(monkeypatch):
    """The default fp32 arithmetic (fp16 hi / lo pairs, three products on the fp16 matrix cores), the same with the backbone
    forward on producer-written plane pairs (RADET_PAIRS=1) and with o1 / d_o2 of the stride-1 bottleneck blocks stored ONLY
    as pairs (RADET_PAIRS_ONLY=1, round 6), the former one (three bf16 planes, six products) and the native
    fp32 MFMA run the same forward + loss + backward on the headline batch.  Yardstick: the native arithmetic against ITSELF with
    other tiles / split-K factors (launcher heuristics instead of the tune file), i.e. another fp32 summation order --
    ReLU masks and GroupNorm amplify last-bit differences through ~60 layers.  The plane arithmetic must sit within 3x of
    that distance (measured: about the same) and far below the bf16 math mode; the loss triple agrees to 1e-6."""
    import os
    import bench
    from radet_amd.models import build_detector
    from radet_amd.utils import Config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    img, boxes, labels, p2g, pw = bench.make_batch(0, 4, torch.device("cuda"))
    out = {}
    for tag, math, tune, env in (("planes", "fp32", "1", {}), ("pairs", "fp32", "1", {"RADET_PAIRS": "1"}),
                                 ("pairs-only", "fp32", "1", {"RADET_PAIRS_ONLY": "1"}),
                                 # the towers on fp32 tensors (no plane-pair operands): grouped launches / two-stream backward --
                                 # their GroupNorm outputs and gradients are conv operands whose amax slots must be raised
                                 ("p3-off", "fp32", "1", {"RADET_P3": "0"}), ("hybrid", "fp32", "1", {"RADET_TOWER_MODE": "hybrid"}),
                                 ("bf16x6", "fp32", "1", {"RADET_X3": "bf16"}), ("mfma", "fp32-mfma", "1", {}),
                                 ("mfma-heur", "fp32-mfma", "0", {}), ("bf16", "bf16", "1", {})):
        monkeypatch.setenv("RADET_AUTOTUNE", tune)
        for k in ("RADET_PAIRS", "RADET_X3", "RADET_PAIRS_ONLY", "RADET_P3", "RADET_TOWER_MODE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        cfg = Config.fromfile(os.path.join(root, "configs", "bop", "r50_ycbv_pbr.py"))
        cfg.model["pretrained"] = None
        torch.manual_seed(0)
        det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
        rt = det.runtime(math=math)
        assert rt.engine.x3 == (math == "fp32") and rt.engine.pairs == (tag == "pairs") and rt.engine.h2 == (tag in ("planes", "pairs", "pairs-only", "p3-off", "hybrid"))
        assert rt.engine.p3 == (tag in ("planes", "pairs", "pairs-only", "bf16x6"))
        assert rt.engine.po == (tag == "pairs-only") and (not rt.engine.po or sum(blk["po"] for st in rt.engine.stages for blk in st) == 13)
        tg = rt.pack_targets([torch.from_numpy(b) for b in boxes], [torch.from_numpy(l) for l in labels], list(p2g), list(pw))
        rt.forward(img)
        losses = rt.loss(tg).clone()
        rt.backward()
        torch.cuda.synchronize()
        out[tag] = (losses.double().cpu(), rt.flat.grads.double().cpu().clone())
        del det, rt
    ref_l, ref_g = out["mfma"]
    dist = {k: float((g - ref_g).norm() / ref_g.norm()) for k, (_, g) in out.items() if k != "mfma"}
    print("gradient distance to the tuned native-fp32 run:", dist)
    for tag in ("planes", "pairs", "pairs-only", "p3-off", "hybrid", "bf16x6"):
        assert ((out[tag][0] - ref_l).abs() / ref_l.abs()).max() < 1e-6, (tag, out[tag][0], ref_l)
        assert dist[tag] <= 3 * dist["mfma-heur"] + 1e-6, dist
    assert dist["bf16"] > 20 * dist["planes"], dist


# ------------------------------------------------------------------------------------------------ BASELINE configs 3 / 5 at full size
def _synth_batch(B, H, W, seed):
    """images + 1..8 boxes with elliptical visible masks per image + GPU-assigned points, like bench.make_batch"""
    import bench
    from radet_amd.datasets import LabelAssignment
    rng = np.random.RandomState(seed)
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, 3, H, W, generator=g).cuda()
    boxes, labels, masks, rngs = [], [], [], []
    for i in range(B):
        b, l, m = bench.synth_objects(rng, int(rng.randint(1, 9)), H, W)
        boxes.append(b); labels.append(l); masks.append(m); rngs.append(np.random.RandomState(seed * 10 + i))
    la = LabelAssignment(neg_threshold=0.2, positive_num=10, adapt_positive_num=False, balance_sample=True)
    p2g, pw = la.assign_batch(boxes, masks, (H, W, 3), rngs=rngs)
    return img, boxes, labels, p2g, pw


def _runtime(depth, math, seed=0):
    import os
    from radet_amd.models import build_detector
    from radet_amd.utils import Config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "bop", "r50_ycbv_pbr.py"))
    cfg.model["pretrained"] = None
    cfg.model["backbone"]["depth"] = depth
    torch.manual_seed(seed)
    det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
    rt = det.runtime(math=math)
    rt.init_optimizer()
    rt.set_loss_from_head(det.bbox_head)
    return det, rt


def test_config5_r101_800x800_bs2_full_size():
    """BASELINE configs[4]: ResNet-101, 800 x 800, bs 2 (levels 100^2 ... 7^2, N = 13 343 points / image, 26 686 head rows):
    the tile / split-K / tail-split choices of this geometry are driven through a whole train step -- finite losses,
    a loss that goes down, two runs bit-identical; the 100 x 100-level GEMM of this geometry is exactly linear under
    power-of-two scalings (fwd / dgrad / wgrad), whatever tile the tuner picked for it."""
    from radet_amd import kernels as K
    outs = []
    for rep in range(2):
        det, rt = _runtime(101, "fp32")
        img, boxes, labels, p2g, pw = _synth_batch(2, 800, 800, seed=7)
        assert p2g.shape == (2, 13343)
        tg = rt.pack_targets([torch.from_numpy(b) for b in boxes], [torch.from_numpy(l) for l in labels], list(p2g), list(pw))
        hist = [rt.train_step(img, tg, lr=1e-4).clone() for _ in range(4)]
        assert rt.engine.R == 26686 and all(torch.isfinite(h).all() for h in hist)
        assert float(hist[-1].sum()) < float(hist[0].sum())
        outs.append((torch.stack(hist).cpu(), rt.flat.params.clone().cpu()))
        if rep == 0:
            e = rt.engine
            g = e.cls_tower[0].geom                                    # all five levels of the 800 x 800 pyramid, B = 2
            gen = torch.Generator().manual_seed(1)
            x = torch.randn(g.lin.rows, 256, generator=gen).cuda()
            w = (torch.randn(256, 9, 256, generator=gen) * 0.02).cuda()
            y1, y2 = torch.empty(g.lout.rows, 256, device="cuda"), torch.empty(g.lout.rows, 256, device="cuda")
            # the tower GEMMs take plane operands by default (engine.p3): same property, operands in the towers' format
            op = (lambda t: K.Planes.from_float(t.reshape(-1, 256))) if e.p3 else (lambda t: t)
            K.conv_fwd(g, op(x), op(w), None, y1, tile=e._ttile(e.cls_tower[0], tag=False))
            K.conv_fwd(g, op(2 * x), op(w / 4), None, y2, tile=e._ttile(e.cls_tower[0], tag=False))
            assert torch.equal(y2 * 2, y1)
            K.conv_dgrad(g, op(x), op(w), y1, tile=e._ttile(e.cls_tower[0], bwd=True, tag=False, pair=False))
            K.conv_dgrad(g, op(x * 0.5), op(w * 8), y2, tile=e._ttile(e.cls_tower[0], bwd=True, tag=False, pair=False))
            assert torch.equal(y2 * 0.25, y1)
            S = g.nsplit
            s1, s2 = torch.empty(S, 256, 9, 256, device="cuda"), torch.empty(S, 256, 9, 256, device="cuda")
            K.conv_wgrad(g, op(y1), op(x), s1)
            K.conv_wgrad(g, op(y1 * 4), op(x * 0.5), s2)
            assert torch.equal(s2 * 0.5, s1)
            lg = e.stages[2][5]["c2"].geom                             # a layer3 3x3 (50 x 50, M = 5000: split-K territory)
            x3 = torch.randn(lg.lin.rows, lg.cin, generator=gen).cuda()
            w3 = (torch.randn(lg.cout, 9, lg.cin, generator=gen) * 0.02).cuda()
            a1, a2 = torch.empty(lg.lout.rows, lg.cout, device="cuda"), torch.empty(lg.lout.rows, lg.cout, device="cuda")
            K.conv_fwd(lg, x3, w3, None, a1, relu=True)
            K.conv_fwd(lg, x3 * 2, w3 * 0.5, None, a2, relu=True)
            assert torch.equal(a1, a2)
        del det, rt
        torch.cuda.empty_cache()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_config3_bf16_storage_640x480_bs4_full_size():
    """BASELINE configs[2] arithmetic at the headline size: bf16 tensors in HBM, bs 4, 640 x 480.  Finite losses, two runs
    bit-identical, and the first-step loss triple within the bf16 bound of the fp32 engine on the same weights / batch
    (bf16 has 8 mantissa bits: relative differences of a few 1e-3 after ~60 rounded layers; bound 3e-2)."""
    img, boxes, labels, p2g, pw = _synth_batch(4, 480, 640, seed=3)
    ref = None
    res = {}
    for math in ("fp32", "bf16-storage", "bf16-storage"):
        det, rt = _runtime(50, math)
        tg = rt.pack_targets([torch.from_numpy(b) for b in boxes], [torch.from_numpy(l) for l in labels], list(p2g), list(pw))
        hist = torch.stack([rt.train_step(img, tg, lr=1e-4).clone() for _ in range(3)]).cpu()
        assert torch.isfinite(hist).all()
        res.setdefault(math, []).append((hist, rt.flat.params.clone().cpu()))
        if math == "fp32":
            ref = hist[0]
        else:
            assert rt.engine.buf["P"].dtype == torch.bfloat16 and rt.engine.B == 4
            rel = ((hist[0] - ref).abs() / ref.abs().clamp_min(1e-6)).max().item()
            assert rel < 3e-2, (hist[0], ref)
        del det, rt
        torch.cuda.empty_cache()
    (h1, p1), (h2, p2) = res["bf16-storage"]
    assert torch.equal(h1, h2) and torch.equal(p1, p2)
