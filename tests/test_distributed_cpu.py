"""N > 1 path on CPU: two gloo ranks run the bucketed gradient exchange of the native train step
(radet_amd.runtime.GradReducer / compute_buckets) on CPU arenas."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build_flat():
    from radet_amd.engine import Engine
    from radet_amd.models import build_detector
    from radet_amd.runtime import FlatParams, compute_buckets
    from radet_amd.utils import Config
    cfg = Config.fromfile(os.path.join(REPO, "configs", "bop", "r50_ycbv_pbr.py"))
    cfg.model["pretrained"] = None
    torch.manual_seed(0)
    det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    flat = FlatParams(det, torch.device("cpu"))
    eng = Engine(flat.p, flat.g, depth=50)
    return det, flat, compute_buckets(flat, [c.name for c in eng.convs], [c.trainable for c in eng.convs])


def test_flat_arena_and_buckets():
    det, flat, buckets = _build_flat()
    # layer4 (59.9 MB, 47 % of the gradient bytes) is cut per bottleneck block, last block first; every message >= 15 MB
    # except the one that cannot be larger (layer2, the last of the backward pass)
    assert [b["prefix"] for b in buckets] == ["bbox_head.", "neck.", "backbone.layer4.2.", "backbone.layer4.1.", "backbone.layer4.0.",
                                              "backbone.layer3.", "backbone.layer2."]
    assert [b.get("blocks") for b in buckets[2:5]] == [(3, 2, 2), (3, 1, 1), (3, 0, 0)]
    assert all(4 * (b["arena"][1] - b["arena"][0]) >= 15e6 for b in buckets[:-1])
    os.environ["RADET_SPLIT_BUCKETS"] = "0"
    try:
        assert [b["prefix"] for b in _build_flat()[2]] == ["bbox_head.", "neck.", "backbone.layer4.", "backbone.layer3.", "backbone.layer2."]
    finally:
        del os.environ["RADET_SPLIT_BUCKETS"]
    spans = sorted(b["arena"] for b in buckets)
    assert spans[0][0] == 0
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 <= b0 and b0 - a1 < 4          # disjoint, only alignment padding between buckets
    assert flat.n_train - spans[-1][1] < 4
    covered = sum(e - b for b, e in spans)
    assert covered >= 31933983
    # parameters are views into the arena (load_state_dict keeps them there)
    p = dict(det.named_parameters())["bbox_head.atss_cls.weight"]
    o = flat.offsets["bbox_head.atss_cls.weight"]
    assert p.data_ptr() == flat.params[o:].data_ptr()
    sd = {k: v.clone() + 1 for k, v in det.state_dict().items() if v.is_floating_point()}
    det.load_state_dict(sd, strict=False)
    assert p.data_ptr() == flat.params[o:].data_ptr() and torch.equal(p.data, sd["bbox_head.atss_cls.weight"])
    # conv-table ranges are contiguous and in table order
    assert all(b["convs"][0] < b["convs"][1] for b in buckets)


def _worker(rank, world, port, q, bf16=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from radet_amd.runtime import GradReducer
    _, flat, buckets = _build_flat()
    g = torch.Generator().manual_seed(100 + rank)
    flat.grads.copy_(torch.randn(flat.n_train, generator=g))
    mine = flat.grads.clone()
    red = GradReducer(flat.grads, torch.device("cpu"), bf16=bf16)
    for b in buckets:                      # backward order, asynchronous
        red.bucket_ready(b)
    red.finish()
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    expect = sum(gathered)
    lo = min(b["arena"][0] for b in buckets)
    hi = max(b["arena"][1] for b in buckets)
    if bf16:      # every rank's gradient is rounded to bf16, the sum is formed in bf16: within 2 bf16 steps of the fp32 sum
        f = world / 2.0                    # (at two ranks; the partial sums of a longer reduction round once per rank)
        ok = bool(((flat.grads[lo:hi] - expect[lo:hi]).abs() <= f * 2.0 ** -7 * expect[lo:hi].abs() + f * 2.0 ** -6).all())
    else:           # (another summation order than sum(gathered): a few ulp of the partial sums per rank)
        ok = torch.allclose(flat.grads[lo:hi], expect[lo:hi], rtol=0, atol=1e-6 * world / 2)
    # whatever the order, EVERY rank holds the same bits afterwards (replicas stay identical)
    dist.all_gather(gathered, flat.grads)
    ok = ok and all(torch.equal(gathered[0][lo:hi], t[lo:hi]) for t in gathered)
    # mean applied downstream: grad_div = world in the fused optimiser kernel
    q.put((rank, bool(ok), float((flat.grads[lo:hi] / world - expect[lo:hi] / world).abs().max())))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 8], ids=["2ranks", "8ranks"])
@pytest.mark.parametrize("bf16", [False, True], ids=["fp32-buckets", "bf16-buckets"])
def test_two_rank_gloo_gradient_exchange(bf16, world):
    """(the name dates from the 2-rank version; world 8 = the node BASELINE's multi-GPU configs run on: seven buckets handed
    over in backward order by eight processes, sums equal on every rank, bf16 staging incl.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, bf16)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world)) and all(r[1] for r in res)


# ---------------------------------------------------------------------------------------------- train-step control flow
def _harness(seed):
    """DetectorRuntime with the GPU compute replaced by host stand-ins: train_step / sync_replicas / GradReducer /
    bucket order are the real code under test, forward / loss are no-ops, backward fills the gradient arena with a
    rank-specific pattern and fires the bucket hook in the engine's backward order, optimizer_step is the reference
    formula of the fused kernel (global-norm clip with grad_div, AdamW)."""
    from radet_amd.engine import Engine
    from radet_amd.models import build_detector
    from radet_amd.runtime import DetectorRuntime, FlatParams, compute_buckets
    from radet_amd.utils import Config

    class Harness(DetectorRuntime):
        def __init__(self, det):
            self.dev = torch.device("cpu")
            self.flat = FlatParams(det, self.dev)
            self.engine = Engine(self.flat.p, self.flat.g, depth=50)
            self.engine.losses = torch.zeros(3)
            self.buckets = compute_buckets(self.flat, [c.name for c in self.engine.convs], [c.trainable for c in self.engine.convs])
            self.opt_state, self.step_count, self.reducer = None, 0, None
            self.order = []
            self.sync_replicas()

        def forward(self, img, fold=True):
            pass

        def loss(self, tg, grad_scale=None, **kw):
            return self.engine.losses

        def backward(self, bucket_hook=None, next_img=None):
            g = torch.Generator().manual_seed(1000 + dist.get_rank())
            self.flat.grads.copy_(torch.randn(self.flat.n_train, generator=g) * self.param_mask())
            for b in self.buckets:                       # head -> neck -> layer4 -> layer3 -> layer2
                self.order.append(b["prefix"])
                if bucket_hook is not None:
                    bucket_hook(b)

        def param_mask(self):
            """1 on parameter elements, 0 on the (< 4 element) alignment gaps of the arena, which no kernel ever writes"""
            m = torch.zeros(self.flat.n_train)
            for n, o in self.flat.offsets.items():
                m[o:o + self.flat.p[n].numel()] = 1
            return m

        def init_optimizer(self, lr=4e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05, max_norm=35.0):
            n = self.flat.n_train
            self.opt_state = dict(m=torch.zeros(n), v=torch.zeros(n), lr=lr, betas=betas, eps=eps, wd=weight_decay, max_norm=max_norm)

        def optimizer_step(self, lr=None, grad_div=1.0):
            st = self.opt_state
            assert not self.reducer.works, "all-reduces must be finished before the optimizer runs"
            self.step_count += 1
            g = self.flat.grads / grad_div
            norm = g.double().norm().float()
            g = g * torch.clamp(st["max_norm"] / (norm + 1e-6), max=1.0)
            b1, b2 = st["betas"]
            st["m"].mul_(b1).add_(g, alpha=1 - b1)
            st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
            p = self.flat.params
            p.mul_(1 - st["lr"] * st["wd"])
            denom = (st["v"] / (1 - b2 ** self.step_count)).sqrt().add_(st["eps"])
            p.addcdiv_(st["m"] / (1 - b1 ** self.step_count), denom, value=-st["lr"])
            self.grad_div = grad_div

    cfg = Config.fromfile(os.path.join(REPO, "configs", "bop", "r50_ycbv_pbr.py"))
    cfg.model["pretrained"] = None
    torch.manual_seed(seed)                               # DIFFERENT initial weights per rank
    det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    return det, Harness(det)


def _step_worker(rank, world, port, q, ckpt):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from radet_amd.apis import load_checkpoint, save_checkpoint
    det, rt = _harness(seed=10 + rank)
    gathered = [torch.zeros_like(rt.flat.params) for _ in range(world)]
    dist.all_gather(gathered, rt.flat.params)
    same_start = all(torch.equal(gathered[0], t) for t in gathered)           # sync_replicas() at construction
    fro = [torch.zeros_like(rt.flat.frozen) for _ in range(world)]
    dist.all_gather(fro, rt.flat.frozen)
    same_start = same_start and all(torch.equal(fro[0], t) for t in fro)
    rt.init_optimizer(max_norm=35.0)
    p0 = rt.flat.params.clone()
    for _ in range(2):
        rt.train_step(None, None, lr=None)
    # expected: AdamW on the MEAN of the ranks' gradients (same generator seeds as Harness.backward)
    mean = sum(torch.randn(rt.flat.n_train, generator=torch.Generator().manual_seed(1000 + r)) for r in range(world)) / world
    mean = mean * rt.param_mask()
    summed_ok = torch.allclose(rt.flat.grads, mean * world, rtol=0, atol=1e-5)
    dist.all_gather(gathered, rt.flat.params)
    same_end = all(torch.equal(gathered[0], t) for t in gathered)
    moved = float((rt.flat.params - p0).abs().max()) > 0
    order_ok = rt.order[:7] == ["bbox_head.", "neck.", "backbone.layer4.2.", "backbone.layer4.1.", "backbone.layer4.0.",
                                "backbone.layer3.", "backbone.layer2."]
    # checkpoint written by rank 0 only, loaded by rank 0 only -> load_checkpoint's broadcast keeps the replicas equal
    if rank == 0:
        save_checkpoint(det, ckpt, meta=dict(iter=2), runtime=rt)
    dist.barrier()
    det2, rt2 = _harness(seed=50 + rank)
    rt2.init_optimizer(max_norm=35.0)
    if rank == 0:
        sd = torch.load(ckpt, map_location="cpu")
        det2.load_state_dict(sd["state_dict"])
        from radet_amd.apis.train import load_optimizer_state_dict
        load_optimizer_state_dict(det2, rt2, sd["optimizer"])
    rt2.sync_replicas()
    resumed = torch.equal(rt2.flat.params, rt.flat.params) and torch.equal(rt2.opt_state["m"], rt.opt_state["m"]) \
        and torch.equal(rt2.opt_state["v"], rt.opt_state["v"]) and rt2.step_count == 2
    q.put((rank, bool(same_start), bool(summed_ok), bool(same_end), bool(moved), bool(order_ok), rt.grad_div == float(world),
           bool(resumed)))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 8], ids=["2ranks", "8ranks"])
def test_two_rank_train_step_control_flow(tmp_path, world):
    """Ranks start from different seeds; after construction (initial broadcast) and two data-parallel steps they hold
    identical parameters; the optimizer saw the summed gradient with grad_div = world, after every all-reduce had
    finished; a checkpoint loaded on rank 0 only reaches every rank through sync_replicas().  World 8 = one process per GPU
    of the node the multi-GPU configs of BASELINE.json run on (buckets, grad_div = 8, replica start and resume at that size)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_step_worker, args=(r, world, port, q, str(tmp_path / "ck.pth"))) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=800) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    for r in res:
        assert all(r[1:]), r


def _parse_losses_of(losses):
    from radet_amd.models.radet import RADet
    return RADet._parse_losses(None, losses)          # the method does not touch the instance


def test_parse_losses_single_process():
    """detectors/base.py:185-216: tensors are averaged, lists of tensors summed entry by entry, the total sums every key that
    contains 'loss' (left to right), other keys are logged only; anything else raises TypeError."""
    a = torch.tensor([1.0, 3.0], requires_grad=True)
    losses = {"loss_cls": a, "loss_bbox": [torch.tensor([2.0, 4.0]), torch.tensor(0.5)], "acc": torch.tensor([10.0, 30.0])}
    loss, log = _parse_losses_of(losses)
    assert list(log) == ["loss_cls", "loss_bbox", "acc", "loss"]
    assert log == {"loss_cls": 2.0, "loss_bbox": 3.5, "acc": 20.0, "loss": 5.5}
    assert all(type(v) is float for v in log.values())
    loss.backward()                                    # the returned loss is differentiable, the log values are detached
    assert torch.equal(a.grad, torch.tensor([0.5, 0.5]))
    with pytest.raises(TypeError, match="loss_x is not a tensor or list of tensors"):
        _parse_losses_of({"loss_x": 1.0})


def _parse_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []
    orig = dist.all_reduce
    dist.all_reduce = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    loss, log = _parse_losses_of({"loss_cls": torch.tensor([1.0 + rank, 3.0 + rank]), "loss_iou": [torch.tensor(4.0 * rank)]})
    dist.all_reduce = orig
    q.put((rank, float(loss), dict(log), len(calls)))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_parse_losses_two_ranks_one_collective():
    """log_vars are the mean over the ranks (as in the reference), obtained with ONE all-reduce of the stacked scalars;
    the loss that is back-propagated stays the local one."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_parse_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=240) for _ in range(world))
    [p.join(30) for p in ps]
    for rank, loss, log, ncalls in res:
        assert ncalls == 1
        assert loss == (2.0 + rank) + 4.0 * rank                 # local
        assert log == {"loss_cls": 2.5, "loss_iou": 2.0, "loss": 4.5}   # mean over the two ranks
