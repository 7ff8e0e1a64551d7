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
    assert [b["prefix"] for b in buckets] == ["bbox_head.", "neck.", "backbone.layer4.", "backbone.layer3.", "backbone.layer2."]
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
        ok = bool(((flat.grads[lo:hi] - expect[lo:hi]).abs() <= 2.0 ** -7 * expect[lo:hi].abs() + 2.0 ** -6).all())
    else:
        ok = torch.allclose(flat.grads[lo:hi], expect[lo:hi], rtol=0, atol=1e-6)
    # mean applied downstream: grad_div = world in the fused optimiser kernel
    q.put((rank, bool(ok), float((flat.grads[lo:hi] / world - expect[lo:hi] / world).abs().max())))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("bf16", [False, True], ids=["fp32-buckets", "bf16-buckets"])
def test_two_rank_gloo_gradient_exchange(bf16):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, bf16)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1] and all(r[1] for r in res)
