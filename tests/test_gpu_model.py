"""Whole-detector parity on the GPU against the reference's own outputs (tests/golden/model.npz):
drop-in config -> build_detector -> forward_train / backward / simple_test."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_det():
    from oracle import synth
    from radet_amd.models import build_detector
    from radet_amd.utils import Config
    cfg = Config.fromfile(os.path.join(REPO, "configs", "bop", "r50_ycbv_pbr.py"))
    cfg.model["pretrained"] = None
    d = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    synth.fill_state_dict(d.state_dict(), seed=0)
    return d.cuda()


@pytest.fixture(scope="module")
def det():
    return make_det()


def targets(golden, tags=("g8", "g3")):
    a = golden("assigner")
    return ([torch.from_numpy(a[t + "_boxes"]) for t in tags], [torch.from_numpy(a[t + "_labels"]) for t in tags],
            [torch.from_numpy(a[t + "_p2g"].astype(np.int64)) for t in tags], [torch.from_numpy(a[t + "_w"]) for t in tags])


def nchw_flat(ts):
    return torch.cat([t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]) for t in ts])


def test_features_and_head_outputs(det, golden):
    from oracle import synth
    g = golden("model")
    img = synth.synth_images(0, 2).cuda()
    det.eval()
    cf = det.backbone(img)
    for i, f in enumerate(cf):
        v = f.reshape(-1)[torch.from_numpy(g[f"c{i + 2}_idx"]).cuda()].cpu().numpy()
        assert np.allclose(v, g[f"c{i + 2}_val"], rtol=1e-4, atol=1e-4 * float(g[f"c{i + 2}_absmean"])), f"C{i + 2}"
    pf = det.extract_feat(img)
    for i, f in enumerate(pf):
        v = f.reshape(-1)[torch.from_numpy(g[f"p{i + 3}_idx"]).cuda()].cpu().numpy()
        assert np.allclose(v, g[f"p{i + 3}_val"], rtol=1e-4, atol=1e-4 * float(g[f"p{i + 3}_absmean"])), f"P{i + 3}"
    outs = det.bbox_head(pf)
    for nm, ts in zip(["cls", "reg", "iou"], outs):
        v = nchw_flat(ts).reshape(-1)[torch.from_numpy(g[f"{nm}_idx"]).cuda()].cpu().numpy()
        assert np.allclose(v, g[f"{nm}_val"], rtol=1e-4, atol=1e-4 * float(g[f"{nm}_absmean"])), nm


def test_train_forward_backward(det, golden):
    from oracle import synth
    g = golden("model")
    img = synth.synth_images(0, 2).cuda()
    gt_b, gt_l, p2g, pw = targets(golden)
    det.train()
    det.zero_grad()
    losses = det(img=img, img_metas=synth.img_metas(2), return_loss=True, gt_bboxes=gt_b, gt_labels=gt_l,
                 points_to_gt_index=p2g, points_weight=pw)
    for k in ("loss_cls", "loss_bbox", "loss_iou"):
        assert abs(losses[k].item() - float(g[k])) <= 1e-4 * max(1.0, abs(float(g[k]))), (k, losses[k].item(), float(g[k]))
    loss, log_vars = det._parse_losses(losses)
    assert abs(log_vars["loss"] - float(g["loss"])) <= 1e-4 * float(g["loss"])
    loss.backward()
    named = dict(det.named_parameters())
    names = [str(n) for n in g["grad_names"]]
    assert sorted(names) == sorted(n for n, p in named.items() if p.requires_grad)
    mine = np.array([named[n].grad.double().norm().item() for n in names])
    ref = g["grad_norms"]
    bad = [(n, a, b) for n, a, b in zip(names, mine, ref) if abs(a - b) > 5e-4 * b + 1e-6 * float(g["total_grad_norm"])]
    assert not bad, bad[:10]
    assert all(named[n].grad is None for n in named if not named[n].requires_grad)
    # the gradient tensors themselves against elements sampled from the reference's gradients (model_grads.npz)
    from _grads import assert_sampled_grads
    assert_sampled_grads({n: named[n].grad for n in names}, golden("model_grads"), atol_total=2e-6,
                         total=float(g["total_grad_norm"]))      # (floor: see tests/_grads.py::assert_grads_close)


def test_simple_test(det, golden):
    from oracle import synth
    g = golden("model")
    img = synth.synth_images(0, 2).cuda()
    det.eval()
    with torch.no_grad():
        results = det(img=[img], img_metas=[synth.img_metas(2)], return_loss=False, rescale=True)
    assert len(results) == 2 and all(len(r) == 21 for r in results)
    for i, per_cls in enumerate(results):
        dets = np.concatenate([np.concatenate([d, np.full((d.shape[0], 1), c, np.float32)], 1) for c, d in enumerate(per_cls)], 0)
        ref = g[f"det_{i}"]
        assert dets.shape == ref.shape
        o, r = np.argsort(-dets[:, 4], kind="stable"), np.argsort(-ref[:, 4], kind="stable")
        assert np.array_equal(dets[o, 5], ref[r, 5])
        assert np.allclose(dets[o, :5], ref[r, :5], rtol=1e-4, atol=2e-3)


def test_native_train_step_matches_autograd(det, golden):
    """runtime.train_step (no autograd, fused clip+AdamW) == autograd path + torch AdamW on a clone."""
    from oracle import synth
    img = synth.synth_images(0, 2).cuda()
    gt_b, gt_l, p2g, pw = targets(golden)
    a, b = make_det(), make_det()
    a.train(); b.train()
    opt = torch.optim.AdamW(a.parameters(), lr=4e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    rt = b.runtime()
    rt.init_optimizer(lr=4e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05, max_norm=35.0)
    tg = rt.pack_targets(gt_b, gt_l, p2g, pw)
    for it in range(2):
        opt.zero_grad()
        losses = a(img=img, img_metas=synth.img_metas(2), return_loss=True, gt_bboxes=gt_b, gt_labels=gt_l,
                   points_to_gt_index=p2g, points_weight=pw)
        sum(losses.values()).backward()
        torch.nn.utils.clip_grad_norm_([p for p in a.parameters() if p.requires_grad], 35.0)
        opt.step()
        lb = rt.train_step(img, tg)
        # step 0 is exact; after one AdamW step (update = +-lr wherever |g| >> eps) the two runs differ by fp32
        # rounding in the parameters, which the sign-like Adam update amplifies -> compare losses only
        assert np.allclose(lb.cpu().numpy(), [losses[k].item() for k in ("loss_cls", "loss_bbox", "loss_iou")],
                           rtol=1e-6 if it == 0 else 2e-3)
        if it == 0:
            pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
            for n in pa:
                d = (pa[n].detach() - pb[n].detach()).abs().max().item()
                assert d <= 1e-6, (n, d)
            assert abs(rt.opt_state["grad_norm"].item() - float(golden("model")["total_grad_norm"])) < 1e-3 * 1053


def test_stream_budget(det, golden):
    """Training, streamed inference, a captured detect graph and the data-parallel exchange together use the main stream and
    at most three more (shared by every engine of the process): a fifth HIP stream in use costs the step a third of its
    speed on this device ("stream budget")."""
    import numpy as np
    from oracle import synth
    from radet_amd.engine import Engine
    from radet_amd.runtime import GradReducer
    img = synth.synth_images(0, 2).cuda()
    gt_b, gt_l, p2g, pw = targets(golden)
    d = make_det()
    d.train()
    rt = d.runtime()
    rt.init_optimizer()
    rt.train_step(img, rt.pack_targets(gt_b, gt_l, p2g, pw))
    d.eval()
    metas = synth.img_metas(2, img.shape[2], img.shape[3])
    list(rt.detect_stream(((img, metas) for _ in range(3)), d.test_cfg))
    rt.detect_graph(img[:1], metas[:1], d.test_cfg)
    torch.cuda.synchronize()
    dev = torch.cuda.current_device()
    roles = {r for (i, r) in Engine._SHARED_STREAMS if i == dev}
    assert roles <= {"side", "side2", "chain"}, roles
    assert not hasattr(rt, "_post_stream") and not hasattr(rt, "_copy_stream")
    e = rt.engine
    assert e._side() is det.runtime().engine._side() and e._chain_stream() is det.runtime().engine._chain_stream()
    # the exchange runs on the chain stream the detector hands it, not on a stream of its own
    red = GradReducer(torch.ones(8, device="cuda"), torch.device("cuda", dev), comm_stream=e._chain_stream())
    assert red.comm_stream is e._chain_stream()
    # a stream of the CALLER's next to the engine's four (the pinned next-batch upload of a data loader; its measured cost is
    # in INTEGRATION.md, tools/bench_user_stream.py): two steps with the copy in flight produce the same bits as without
    d.train()
    tg = rt.pack_targets(gt_b, gt_l, p2g, pw)
    host, nxt = img.cpu().pin_memory(), torch.empty_like(img)

    def two_steps(stream):
        st = d.state_dict()
        snap = {k: v.clone() for k, v in st.items()}
        m, v = rt.opt_state["m"].clone(), rt.opt_state["v"].clone()
        cnt = rt.step_count
        out = []
        for _ in range(2):
            if stream is not None:
                with torch.cuda.stream(stream):
                    nxt.copy_(host, non_blocking=True)
            out.append(rt.train_step(img, tg).clone())
        torch.cuda.synchronize()
        res = (torch.stack(out).cpu(), rt.flat.params.clone())
        with torch.no_grad():                       # rewind
            for k, t in st.items():
                t.copy_(snap[k])
        rt.opt_state["m"].copy_(m); rt.opt_state["v"].copy_(v); rt.step_count = cnt
        rt.engine.params_changed()
        return res
    base = two_steps(None)
    for stream in (torch.cuda.Stream(),):
        got = two_steps(stream)
        assert torch.equal(got[0], base[0]) and torch.equal(got[1], base[1])
        assert torch.equal(nxt, img)


_OWN_STREAM_PROBE = r"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
import radet_amd                                   # (sets GPU_MAX_HW_QUEUES before the first HIP call)
from radet_amd.runtime import GradReducer
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
g = torch.ones(1 << 20, device="cuda")
comm = torch.cuda.Stream() if os.environ.get("PROBE_COMM_STREAM") == "1" else None      # the detector passes its chain stream
red = GradReducer(g, torch.device("cuda", 0), comm_stream=comm)
side, probe = torch.cuda.Stream(), torch.cuda.Stream()
main_ev, side_ev, done_ev = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
with torch.cuda.stream(side):                      # first use: communicator / stream set-up is not part of the probe
    red.bucket_ready(dict(prefix="warm", arena=(0, g.numel())))
torch.cuda.synchronize(); red.finish(); torch.cuda.synchronize()
torch.cuda._sleep(int(2e9))                        # ~1 s on the main stream, BEFORE the hand-over
main_ev.record()
with torch.cuda.stream(side):
    red.bucket_ready(dict(prefix="probe", arena=(0, g.numel())))
    torch.cuda._sleep(int(2e9))                    # the handing stream parked AFTER the hand-over
    side_ev.record()
with torch.cuda.stream(probe):                     # a third stream that only waits for the collective
    red.works[-1][0].wait()
    done_ev.record()
done_ev.synchronize()                              # host waits for the collective only
print("PARKED", not main_ev.query(), not side_ev.query(), flush=True)
torch.cuda.synchronize(); red.finish()
assert float(g[0]) == 1.0
dist.destroy_process_group()
"""


@pytest.mark.parametrize("comm_stream", ["1", "0"], ids=["on-a-stream-of-ours", "process-group-stream"])
def test_collectives_run_on_their_own_stream(comm_stream):
    """The bucket all-reduces -- synchronous collectives on a stream the caller names (what the detector does: its
    tower-chain stream), or asynchronous ones on the process group's internal stream -- neither wait for the main stream nor
    for what the handing (side) stream does after the hand-over: with the main stream parked in a long spin kernel BEFORE the hand-over and the side stream parked in one AFTER
    it, the collective still completes at once.  Run in a process of its own: in the test process dozens of streams from the
    other tests share the device's hardware queues, and a collective queued behind a parked stream's spin kernel in the same
    hardware queue says nothing about stream dependencies."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _OWN_STREAM_PROBE], cwd=root, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, PROBE_COMM_STREAM=comm_stream))
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("PARKED")][-1]
    assert line.split()[1:] == ["True", "True"], line


def test_rccl_bucket_exchange_single_rank(det, golden):
    """The data-parallel step with a 1-rank RCCL group: bucketed async all-reduce on the comm stream, unfold on the
    side stream, mean folded into AdamW -- must equal the plain single-GPU step bit for bit."""
    import torch.distributed as dist
    from oracle import synth
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        img = synth.synth_images(0, 2).cuda()
        gt_b, gt_l, p2g, pw = targets(golden)
        a, b = make_det(), make_det()
        a.train(); b.train()
        ra, rb = a.runtime(), b.runtime()
        for rt in (ra, rb):
            rt.init_optimizer()
        tg = ra.pack_targets(gt_b, gt_l, p2g, pw)
        os.environ["RADET_FORCE_REDUCER"] = "1"
        la = ra.train_step(img, tg).clone()
        os.environ["RADET_FORCE_REDUCER"] = "0"
        lb = rb.train_step(img, tg).clone()
        torch.cuda.synchronize()
        assert torch.equal(la, lb)
        assert ra.reducer is not None and rb.reducer is None
        assert torch.equal(ra.flat.grads, rb.flat.grads) and torch.equal(ra.flat.params, rb.flat.params)
        # the traced step: every bucket handed over during the backward pass and completed; report well-formed
        ra.reducer.enable_trace(True)
        os.environ["RADET_FORCE_REDUCER"] = "1"
        ra.train_step(img, tg)
        rep = ra.comm_report()
        assert [b["bucket"] for b in rep["buckets"]] == [b["prefix"] for b in ra.buckets]
        assert all(0.0 <= b["ready_ms"] <= b["done_by_ms"] for b in rep["buckets"]) and rep["exposed_comm_ms"] >= 0.0
        assert rep["steps"] == 1 and rep["backward_ms"] > 0.0
        assert abs(sum(b["mbytes"] for b in rep["buckets"]) - ra.flat.n_train * 4 / 1e6) < 1.0
    finally:
        os.environ["RADET_FORCE_REDUCER"] = "0"
        dist.destroy_process_group()


def test_train_harness_and_checkpoint(golden, tmp_path):
    """apis.train_detector: OneCycle lr + native steps reduce the loss on a fixed batch; checkpoint restores
    parameters and AdamW state exactly (the next step after a reload equals the uninterrupted one)."""
    from oracle import synth
    from radet_amd.apis import inference_detector, load_checkpoint, save_checkpoint, train_detector
    from radet_amd.utils import Config
    cfg = Config.fromfile(os.path.join(REPO, "configs", "bop", "r50_ycbv_pbr.py"))
    cfg.merge_from_dict({"lr_config.total_steps": 40, "log_config.interval": 5})
    gt_b, gt_l, p2g, pw = targets(golden)
    batch = dict(img=synth.synth_images(0, 2), gt_bboxes=gt_b, gt_labels=gt_l, points_to_gt_index=p2g, points_weight=pw)
    det = make_det()
    logs = []
    hist = train_detector(det, [batch] * 20, cfg, max_iters=20, log=logs.append)
    assert len(hist) == 4 and all(np.isfinite(h).all() for h in hist) and "lr:" in logs[0]
    assert sum(hist[-1]) < sum(hist[0])
    rt = det.runtime()
    path = save_checkpoint(det, str(tmp_path / "ck.pth"), meta=dict(iter=20), runtime=rt)
    tg = rt.pack_targets(gt_b, gt_l, p2g, pw)
    ref_losses = rt.train_step(batch["img"].cuda(), tg, lr=1e-4).clone()
    ref_params = rt.flat.params.clone()
    other = make_det()
    ro = other.train().runtime()
    ro.init_optimizer(max_norm=35.0)
    ro.loss_hparams = rt.loss_hparams
    meta, _ = load_checkpoint(other, path, strict=True, runtime=ro)
    assert meta["iter"] == 20 and ro.step_count == 20
    lo = ro.train_step(batch["img"].cuda(), ro.pack_targets(gt_b, gt_l, p2g, pw), lr=1e-4)
    assert torch.equal(lo, ref_losses) and torch.equal(ro.flat.params, ref_params)
    res = inference_detector(other.eval(), batch["img"].cuda())
    assert len(res) == 2 and len(res[0]) == 21 and all(r.shape[1] == 5 for r in res[0])


def test_detect_hipgraph_replay_matches_eager():
    """rt.detect_graph captures fold + forward (two streams) + decode + the NMS pipeline in one hipGraph; replays with
    new inputs must equal the eager path bit for bit."""
    import numpy as np
    det = make_det().eval()
    rt = det.runtime()
    with torch.no_grad():
        det.bbox_head.atss_cls.bias += 2.0
    metas = [dict(img_shape=(160, 192, 3), scale_factor=np.ones(4, np.float32)) for _ in range(2)]
    g = torch.Generator().manual_seed(3)
    for rep in range(3):
        img = torch.randn(2, 3, 160, 192, generator=g).cuda()
        a = rt.detect(img, metas, det.test_cfg, rescale=True)
        b = rt.detect_graph(img, metas, det.test_cfg, rescale=True)
        assert len(a) == len(b) == 2
        for (da, la), (db, lb) in zip(a, b):
            assert torch.equal(da, db) and torch.equal(la, lb)
        assert sum(d.shape[0] for d, _ in a) > 0


def test_folded_weights_are_cached_until_parameters_change(monkeypatch):
    """Engine.fold() keeps the folded (BN-scaled, re-laid-out) weights while their sources are unchanged: inference calls
    fold nothing after the first, a train step re-folds only the trainable convs, and any write through torch (here an
    in-place scale of a frozen stem weight / of a trainable head weight) or `invalidate_fold()` is picked up; results
    equal those of folding on every call."""
    from radet_amd import kernels as K
    det = make_det().eval()
    rt = det.runtime()
    e = rt.engine
    calls = []
    real = K.fold_weights
    monkeypatch.setattr(K, "fold_weights", lambda table, n: (calls.append(n), real(table, n))[1])
    img = torch.randn(2, 3, 160, 192, generator=torch.Generator().manual_seed(5)).cuda()
    with torch.no_grad():
        f0 = [t.clone() for t in det.extract_feat(img)]
        n_all = sum(calls)
        assert n_all == len(e.convs)
        calls.clear()
        f1 = det.extract_feat(img)
        assert calls == [] and all(torch.equal(a, b) for a, b in zip(f0, f1))
        e.p["backbone.conv1.weight"].mul_(1.25)                   # frozen stem, written through the arena view
        f2 = [t.clone() for t in det.extract_feat(img)]
        nf = sum(1 for c in e.convs if not c.trainable)           # leading frozen convs (stem + layer1)
        assert 0 < nf < n_all and sum(calls) >= nf and not torch.equal(f2[0], f0[0])
        calls.clear()
        det.bbox_head.atss_cls.weight.mul_(0.5)                   # trainable, written through the module's Parameter
        f2 = [t.clone() for t in det.extract_feat(img)]
        assert sum(calls) >= n_all - nf
        monkeypatch.setenv("RADET_FOLD_EVERY_CALL", "1")
        calls.clear()
        f3 = det.extract_feat(img)
        assert sum(calls) == n_all and all(torch.equal(a, b) for a, b in zip(f2, f3))
        monkeypatch.delenv("RADET_FOLD_EVERY_CALL")
        e.invalidate_fold()
        det.extract_feat(img)
        calls.clear()
        e.params_changed()                                        # what the fused AdamW step reports
        det.extract_feat(img)
        assert sum(calls) == n_all - nf and nf > 0                # trainable part only


def test_multi_geometry_plans_no_retune_no_realloc():
    """Engine keeps a plan per (B, H, W): alternating training at B = 4 with detection at B = 1 (a harness that
    interleaves validation) builds each plan once -- no second autotune, no reallocation, identical results."""
    from oracle import synth
    from radet_amd import kernels as K
    det = make_det().train()
    rt = det.runtime()
    rt.init_optimizer(max_norm=35.0)
    rt.set_loss_from_head(det.bbox_head)
    img4 = synth.synth_images(3, 4).cuda()
    img1 = synth.synth_images(4, 1).cuda()
    a = np.load(os.path.join(os.path.dirname(__file__), "golden", "assigner.npz"))
    tags = ("g8", "g3", "g1", "g20")
    tg = rt.pack_targets([torch.from_numpy(a[t + "_boxes"]) for t in tags], [torch.from_numpy(a[t + "_labels"]) for t in tags],
                         [torch.from_numpy(a[t + "_p2g"].astype(np.int64)) for t in tags], [torch.from_numpy(a[t + "_w"]) for t in tags])
    metas = synth.img_metas(1)
    rt.train_step(img4, tg, lr=1e-5)
    first = rt.detect(img1, metas, det.test_cfg)
    torch.cuda.synchronize()
    built, tuned = rt.engine.plans_built, K.TUNE_RUNS
    ptr4 = None
    mem = None
    for it in range(10):
        rt.train_step(img4, tg, lr=0.0)                       # lr = 0 and wd * lr = 0: weights stay put
        p = rt.engine.buf["P"].data_ptr()
        assert ptr4 is None or p == ptr4
        ptr4 = p
        res = rt.detect(img1, metas, det.test_cfg)
        assert rt.engine.buf["P"].data_ptr() != ptr4           # the B = 1 plan has its own buffers
        torch.cuda.synchronize()
        if it == 1:
            mem = torch.cuda.memory_allocated()
        if it > 1:
            assert torch.cuda.memory_allocated() == mem
    assert rt.engine.plans_built == built == 2 and K.TUNE_RUNS == tuned
    assert torch.equal(res[0][0], first[0][0]) and torch.equal(res[0][1], first[0][1])
    # LRU bound: a 5th geometry evicts the oldest plan, which is rebuilt (not corrupted) when it comes back
    for hw in ((128, 160), (160, 128), (96, 128), (128, 96)):
        rt.detect(synth.synth_images(5, 1, *hw).cuda(), synth.img_metas(1, *hw), det.test_cfg)
    assert len(rt.engine._plans) <= rt.engine.max_plans
    again = rt.detect(img1, metas, det.test_cfg)
    assert torch.equal(again[0][0], first[0][0])


def test_detect_stream_abandoned_mid_iteration():
    """A caller that breaks out of rt.detect_stream leaves a batch's decode / NMS in flight on the chain stream, reading the
    head outputs and the shared post-processing workspaces: the generator's clean-up makes the main stream wait for it and
    hands the plan its own buffer set back, so the next detect() neither races with it nor sees swapped buffers."""
    import numpy as np
    det = make_det().eval()
    rt = det.runtime()
    with torch.no_grad():
        det.bbox_head.atss_cls.bias += 2.0
    g = torch.Generator().manual_seed(9)
    metas = [dict(img_shape=(320, 384, 3), scale_factor=np.ones(4, np.float32)) for _ in range(2)]
    work = [(torch.randn(2, 3, 320, 384, generator=g).cuda(), metas) for _ in range(4)]
    ref = [rt.detect(im, m, det.test_cfg, rescale=True) for im, m in work]
    own = {k: rt.engine.buf[k].data_ptr() for k in ("cls", "reg_u", "iou")}
    for stop_after in (1, 2):
        it = rt.detect_stream(iter(work), det.test_cfg, rescale=True)
        for k, got in enumerate(it):
            for (da, la), (db, lb) in zip(got, ref[k]):
                assert torch.equal(da, db) and torch.equal(la, lb)
            if k + 1 == stop_after:
                break
        it.close()                                   # (what leaving the loop does once the generator is collected)
        assert own == {k: rt.engine.buf[k].data_ptr() for k in own}
        again = rt.detect(*work[3], det.test_cfg, rescale=True)     # at once, on the main stream
        for (da, la), (db, lb) in zip(again, ref[3]):
            assert torch.equal(da, db) and torch.equal(la, lb) and da.shape[0] > 0


def test_detect_stream_matches_detect():
    """rt.detect_stream (host one batch behind the device) yields the same detections as rt.detect, batch by batch."""
    import numpy as np
    det = make_det().eval()
    rt = det.runtime()
    with torch.no_grad():
        det.bbox_head.atss_cls.bias += 2.0
    g = torch.Generator().manual_seed(4)
    # two geometries in one stream (two plans, two pairs of head-output sets), an odd number of batches each; decode + NMS of a
    # batch run next to the forward pass of the next one, on their own stream
    work = []
    for (h, w, n) in ((160, 192, 3), (320, 384, 5)):
        metas = [dict(img_shape=(h, w, 3), scale_factor=np.full(4, 0.5 + 0.25 * i, np.float32)) for i in range(2)]
        work += [(torch.randn(2, 3, h, w, generator=g).cuda(), metas) for _ in range(n)]
    ref = [rt.detect(im, metas, det.test_cfg, rescale=True) for im, metas in work]
    # a graph captured BEFORE the streams holds pointers into the plan's head-output buffers: the streams (odd batch counts)
    # must leave the plan with its own set, and the graph's buffers alive
    g_im, g_metas = work[-1]
    g_ref = rt.detect_graph(g_im, g_metas, det.test_cfg, rescale=True)
    own = {k: rt.engine.buf[k].data_ptr() for k in ("cls", "reg_u", "iou")}
    for _ in range(2):
        got = list(rt.detect_stream(iter(work), det.test_cfg, rescale=True))
        assert len(got) == len(ref)
        for a, b in zip(got, ref):
            for (da, la), (db, lb) in zip(a, b):
                assert torch.equal(da, db) and torch.equal(la, lb) and da.shape[0] > 0
        assert own == {k: rt.engine.buf[k].data_ptr() for k in own}
        again = rt.detect_graph(g_im, g_metas, det.test_cfg, rescale=True)
        for (da, la), (db, lb) in zip(again, g_ref):
            assert torch.equal(da, db) and torch.equal(la, lb)


def _oracle_run(dtype, img, gt_b, gt_l, p2g, pw, relu_hook=None, depth=50):
    """losses + named gradients of the oracle (torch CPU) with parameters and activations in `dtype`; relu_hook: see
    oracle/model.py RELU_HOOK"""
    from oracle import model as om
    odet = om.OracleDetector(depth, seed=0)
    if dtype == torch.float64:
        for k, t in list(odet.sd.items()):
            if t.is_floating_point():
                odet.sd[k] = t.detach().double().requires_grad_(t.requires_grad)
    om.RELU_HOOK = relu_hook
    try:
        losses = odet.forward_train(img.to(dtype), gt_b, gt_l, p2g, pw)
    finally:
        om.RELU_HOOK = None
    om.parse_losses(losses).backward()
    return {k: float(v.detach()) for k, v in losses.items()}, {n: g.detach().double() for n, g in odet.named_grads().items()}


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("case", ["r50_640x480_bs2", "r50_640x480_bs4", "r101_800x800_bs2", "r101_200x264_bs2"])
def test_gradients_vs_fp64_oracle(golden, case):
    """Whole-model gradients of every trainable parameter against the oracle run in fp64, the yardstick being the SAME
    oracle in fp32 (torch CPU) -- at B = 2 640 x 480 (the reference's golden targets), and, round 6, at the FULL sizes of
    BASELINE configs[1] (r50, 640 x 480, bs 4: the bench's batch) and configs[4] (R101, 800 x 800, bs 2), and on the odd
    sizes of tests/test_gpu_configs.py::test_train_step_vs_oracle (R101, 200 x 264: feature maps 25 x 33 ... 2 x 3, tiles the
    tuner picks on the spot).
    A ReLU whose pre-activation lies within rounding of zero is decided differently by two correct fp32 implementations, and
    one flipped mask moves a whole row of a weight gradient (why tests/_grads.py accepts 3e-3 per parameter against an fp32
    reference).  Here the discrete part is taken out: the fp64 oracle runs once with its own masks -- every element where
    the engine decided otherwise is listed and must be a knife edge (|pre-activation| <= 2e-6 of the tensor's largest) -- and
    once with the ENGINE's masks handed in (oracle RELU_HOOK).  Against that run every parameter's gradient must be as close
    as torch-fp32's is to fp64 (<= 1.5 x + 1e-6 of the total gradient norm) and within 1e-4 of its own norm; the losses
    agree to 2e-6."""
    from oracle import synth
    from _grads import grad_rel_errors
    depth = 101 if case.startswith("r101") else 50
    if case == "r50_640x480_bs2":
        B, H, W = 2, 480, 640
        img = synth.synth_images(0, 2)
        gt_b, gt_l, p2g, pw = targets(golden)
    elif "200x264" in case:
        from test_gpu_configs import batch
        B, H, W = 2, 200, 264
        img, gt_b, gt_l, p2g, pw = batch(H, W, B)                 # (one image of it has no gts)
    else:
        from test_gpu_properties import _synth_batch
        B, H, W = (4, 480, 640) if depth == 50 else (2, 800, 800)
        img_d, boxes, labels, p2g_d, pw_d = _synth_batch(B, H, W, seed=7)
        img = img_d.cpu()
        gt_b, gt_l = [torch.from_numpy(b) for b in boxes], [torch.from_numpy(l) for l in labels]
        p2g, pw = [t.cpu().long() for t in p2g_d], [t.cpu().float() for t in pw_d]
    from radet_amd.models import build_detector
    from radet_amd.utils import Config
    cfg = Config.fromfile(os.path.join(REPO, "configs", "bop", "r50_ycbv_pbr.py"))
    cfg.model["pretrained"] = None
    cfg.model["backbone"]["depth"] = depth
    det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    synth.fill_state_dict(det.state_dict(), seed=0)
    det = det.cuda().train()
    det.zero_grad()
    losses = det(img=img.cuda(), img_metas=synth.img_metas(B, H, W), return_loss=True, gt_bboxes=gt_b, gt_labels=gt_l,
                 points_to_gt_index=p2g, points_weight=pw)
    sum(losses.values()).backward()
    mine = {n: p.grad for n, p in det.named_parameters() if p.requires_grad}
    e = det.runtime().engine

    def nchw(rows, h, w):
        return rows.reshape(B, h, w, -1).permute(0, 3, 1, 2)

    def buf(name):
        t = e.buf[name]
        return (t.to_float() if hasattr(t, "to_float") else t).float().cpu()

    scales = e.scales_tensor().float().cpu()
    cache = {}

    def engine_mask(name, x):
        """the engine's decision for ReLU `name` (1 where its stored activation is positive)"""
        h, w = x.shape[2:]
        if name == "stem":
            return None                                   # frozen, feeds the max-pool only
        if name.startswith("l"):
            return nchw(buf(name), h, w) > 0
        kind, lvl = name.rsplit(".L", 1)
        lvl = int(lvl)
        r0, r1 = e.plv.level_rows(lvl)
        if kind == "bbox":
            return nchw(buf("reg_u")[r0:r1] * scales[lvl], h, w) > 0
        # tower layer: the GroupNorm backward kernel recomputes the decision from z and the stored statistics,
        # ((z - mean) * rstd) * gamma + beta <= 0 in fp32 (layers.hip gn_bwd_*): the same expression here
        t, i = kind.split(".y")
        if kind not in cache:
            cache[kind] = (buf(f"{t}.z{i}"), e.buf[f"{t}.stats{i}"].float().cpu().view(-1, B, 32, 2),
                           e.p[f"bbox_head.{t}_convs.{i}.gn.weight"].float().cpu(), e.p[f"bbox_head.{t}_convs.{i}.gn.bias"].float().cpu())
        z, st, gam, bet = cache[kind]
        zl = z[r0:r1].view(B, h * w, 32, 8)
        mean, rstd = st[lvl, :, :, 0].view(B, 1, 32, 1), st[lvl, :, :, 1].view(B, 1, 32, 1)
        y = ((zl - mean) * rstd) * gam.view(1, 1, 32, 8) + bet.view(1, 1, 32, 8)
        return nchw(y.reshape(B * h * w, 256), h, w) > 0

    flips = []

    def hook_record(name, x):                             # own masks; records where the engine decided differently
        m = engine_mask(name, x)
        if m is not None:
            own = x > 0
            d = own != m
            if bool(d.any()):
                flips.append((name, int(d.sum()), float(x[d].abs().max() / x.abs().max())))
        return None

    l64, g64 = _oracle_run(torch.float64, img, gt_b, gt_l, p2g, pw, relu_hook=hook_record, depth=depth)
    l32, g32 = _oracle_run(torch.float32, img, gt_b, gt_l, p2g, pw, depth=depth)
    l64m, g64m = _oracle_run(torch.float64, img, gt_b, gt_l, p2g, pw, relu_hook=engine_mask, depth=depth)
    n_el = sum(n for _, n, _ in flips)
    print("ReLU decisions that differ from the fp64 oracle's:", flips)
    budget = 64 * max(1.0, B * H * W / (2 * 480 * 640)) * (2 if depth == 101 else 1)      # a handful of knife edges among ~80 M decisions at B = 2
    assert n_el <= budget and all(rel <= 2e-6 for _, _, rel in flips), flips
    for k in ("loss_cls", "loss_bbox", "loss_iou"):
        assert abs(float(losses[k]) - l64[k]) <= 2e-6 * max(1.0, abs(l64[k])), (k, float(losses[k]), l64[k])
    assert sorted(mine) == sorted(g64)
    e_eng, tot = grad_rel_errors(mine, g64m)
    e_f32, _ = grad_rel_errors(g32, g64)
    bad = []
    for n, (d, b) in e_eng.items():
        if d > 1.5 * e_f32[n][0] + 1e-6 * tot or d > 1e-4 * b + 1e-6 * tot:
            bad.append((n, d / max(b, 1e-30), e_f32[n][0] / max(b, 1e-30)))
    assert not bad, sorted(bad, key=lambda t: -t[1])[:10]
    med_e = float(np.median([d / max(b, 1e-30) for d, b in e_eng.values()]))
    med_t = float(np.median([d / max(b, 1e-30) for d, b in e_f32.values()]))
    print(f"{case}: median per-parameter gradient error against fp64: engine {med_e:.2e}, torch fp32 {med_t:.2e}")
    assert med_e <= 1.5 * med_t, (med_e, med_t)


def test_ten_step_trajectory_vs_oracle():
    """Ten optimisation steps on a fixed batch (B = 2, 320 x 256) -- OneCycle learning rate (mmcv defaults,
    default_runtime.py:1-19), global-norm clip at 35, AdamW (lr 4e-4, wd 0.05), apis/train.py:87-169 -- with the fused
    clip + AdamW step against the oracle's trajectories (torch CPU: forward_train + torch.optim.AdamW + clip_grad_norm_) in
    fp32 AND in fp64.  Adam's update is sign-like wherever |g| >> eps and ReLU masks are discrete, so two correct fp32 runs
    drift apart from step to step (measured on this batch: the fp32 oracle leaves the fp64 trajectory by 8e-8, 1e-6, 2e-6,
    4e-5, 3e-4, ... 5e-2 of the loss over the ten steps -- the deviation grows about tenfold per step while the random-init
    losses still move violently; the engine: 2e-7, 7e-7, 8e-7, 8e-6, 1e-4, 2e-3, ... 3e-2).  The yardstick for the engine's
    distance to the fp64 trajectory is therefore the fp32 oracle's own distance to it: per step, the engine's loss triple may
    be off by at most 10 x (one step of that growth) the largest deviation the fp32 oracle has shown up to that step
    (+ 2e-6 relative).  The first step's gradient norm agrees to 1e-4."""
    from oracle import assigner, model as om, synth
    from radet_amd.apis.train import OneCycleLR
    H, W = 256, 320
    img = synth.synth_images(5, 2, H, W)
    boxes = [np.array([[30, 40, 150, 200], [160, 20, 300, 120], [100, 130, 220, 250]], np.float32),
             np.array([[20, 20, 120, 110], [140, 90, 310, 240]], np.float32)]
    labels = [np.array([2, 9, 14], np.int64), np.array([5, 17], np.int64)]
    gt_b, gt_l, p2g, pw = [], [], [], []
    for i, (bx, lb) in enumerate(zip(boxes, labels)):
        masks = np.zeros((len(bx), H, W), np.uint8)
        for k, (x0, y0, x1, y1) in enumerate(bx.astype(int)):
            masks[k, y0:y1, x0:x1] = 1
        a, w_ = assigner.assign_points(bx, lb, masks, (H, W, 3), rng=np.random.RandomState(i))
        gt_b.append(torch.from_numpy(bx)); gt_l.append(torch.from_numpy(lb))
        p2g.append(torch.from_numpy(a)); pw.append(torch.from_numpy(w_))
    sched = OneCycleLR(4e-4, total_steps=40)

    def oracle_traj(dtype):
        odet = om.OracleDetector(50, seed=0)
        if dtype == torch.float64:
            for k, t in list(odet.sd.items()):
                if t.is_floating_point():
                    odet.sd[k] = t.detach().double().requires_grad_(t.requires_grad)
        params = [t for t in odet.sd.values() if t.requires_grad]
        opt = torch.optim.AdamW(params, lr=4e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
        out, norms = [], []
        for it in range(10):
            for g_ in opt.param_groups:
                g_["lr"] = sched.get_lr(it)
            opt.zero_grad(set_to_none=True)
            losses = odet.forward_train(img.to(dtype), gt_b, gt_l, p2g, pw)
            om.parse_losses(losses).backward()
            norms.append(float(torch.nn.utils.clip_grad_norm_(params, 35.0)))
            opt.step()
            out.append([float(losses[k].detach()) for k in ("loss_cls", "loss_bbox", "loss_iou")])
        return np.array(out), norms

    o64, n64 = oracle_traj(torch.float64)
    o32, _ = oracle_traj(torch.float32)
    det = make_det().train()
    rt = det.runtime()
    rt.init_optimizer(lr=4e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05, max_norm=35.0)
    rt.set_loss_from_head(det.bbox_head)
    tg = rt.pack_targets(gt_b, gt_l, p2g, pw)
    gi = img.cuda()
    mine = []
    for it in range(10):
        mine.append(rt.train_step(gi, tg, lr=sched.get_lr(it), next_img=gi).cpu().numpy().tolist())
        if it == 0:
            gn = float(rt.opt_state["grad_norm"])
            assert abs(gn - n64[0]) <= 1e-4 * n64[0], (gn, n64[0])
    mine = np.array(mine)
    scale = np.maximum(1.0, np.abs(o64))
    dev_o = np.maximum.accumulate((np.abs(o32 - o64) / scale).max(1))          # the fp32 oracle's drift, running maximum
    dev_e = (np.abs(mine - o64) / scale).max(1)
    print("relative deviation from the fp64 trajectory per step: fp32 oracle", dev_o, "engine", dev_e)
    assert (dev_e <= 10.0 * dev_o + 2e-6).all(), (dev_e, dev_o)
    assert o64[-1].sum() < o64[0].sum() and mine[-1].sum() < mine[0].sum()
