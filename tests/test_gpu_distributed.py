"""Data-parallel train step with the real engine and TWO ranks on one MI355X (both processes share cuda:0; the
process group is gloo, because RCCL refuses two ranks on one device -- the collective calls, the bucket hooks on the
side stream, finish() before AdamW and grad_div = world are exactly the code the 8-GPU RCCL run executes)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(seed):
    import sys
    sys.path.insert(0, REPO)
    from radet_amd.models import build_detector
    from radet_amd.utils import Config
    cfg = Config.fromfile(os.path.join(REPO, "configs", "bop", "r50_ycbv_pbr.py"))
    cfg.model["pretrained"] = None
    torch.manual_seed(seed)
    det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
    rt = det.runtime()
    rt.init_optimizer()
    rt.set_loss_from_head(det.bbox_head)
    return det, rt


def _batch(rank, rt):
    import sys
    sys.path.insert(0, REPO)
    import bench
    img, boxes, labels, p2g, pw = bench.make_batch(rank, 2, torch.device("cuda"))
    tg = rt.pack_targets([torch.from_numpy(b) for b in boxes], [torch.from_numpy(l) for l in labels], list(p2g), list(pw))
    return img, tg


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      GPU_MAX_HW_QUEUES="8")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    det, rt = _setup(seed=100 + rank)              # different initial weights: the runtime's broadcast must equalise them
    img, tg = _batch(rank, rt)
    losses = [rt.train_step(img, tg, lr=1e-4).clone().cpu() for _ in range(2)]
    torch.cuda.synchronize()
    torch.save(dict(params=rt.flat.params.cpu(), losses=torch.stack(losses), m=rt.opt_state["m"].cpu()),
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_one_gpu_train_step(tmp_path):
    world = 2
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(800)
        assert p.exitcode == 0
    r0, r1 = (torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) for r in range(2))
    assert torch.equal(r0["params"], r1["params"]) and torch.equal(r0["m"], r1["m"])       # replicas stay identical
    assert torch.isfinite(r0["losses"]).all() and not torch.equal(r0["losses"], r1["losses"])   # different shards
    # single-process emulation of the same two steps: rank 0's initial weights, both shards' gradients summed,
    # mean folded into AdamW (grad_div = 2).  Every reduction is deterministic, so the result is bit-identical.
    det, rt = _setup(seed=100)
    shards = [_batch(r, rt) for r in range(2)]
    for _ in range(2):
        total = torch.zeros_like(rt.flat.grads)
        for img, tg in shards:
            rt.forward(img)
            rt.loss(tg)
            rt.backward()
            torch.cuda.synchronize()
            total += rt.flat.grads
        rt.flat.grads.copy_(total)
        rt.optimizer_step(lr=1e-4, grad_div=2.0)
    torch.cuda.synchronize()
    assert torch.equal(rt.flat.params.cpu(), r0["params"])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("launcher", ["torchrun", "plain"])
def test_bench_script_two_ranks_on_one_gpu(launcher):
    """bench.py's own N > 1 path, launched the way the driver launches it (torch.distributed.run, one process per rank) and as a
    plain `python bench.py --gpus 2` (the script then starts the workers itself, before it touches the GPU),
    with the two ranks sharing cuda:0 over gloo (test hooks RADET_BENCH_SHARE_GPU / RADET_BENCH_BACKEND): the barrier +
    synchronize bracketing, the MAX over ranks, the per-rank step times, the traced bucket exchange (`comm`) and the one JSON
    line on rank 0 -- the code the 8-GPU RCCL run executes, minus RCCL itself."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, RADET_BENCH_SHARE_GPU="1", RADET_BENCH_BACKEND="gloo", GPU_MAX_HW_QUEUES="8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2"]
    if launcher == "plain":
        cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2"]
        env = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=800, env=env, cwd=REPO)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-400:], r.stderr[-1500:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 8 and d["config"]["parallelism"] == "dp2"
    assert abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) <= 0.01 * d["value"]           # whole-job images / s
    rk = d["ranks"]
    assert len(rk["per_rank_ms_per_step"]) == 2 and rk["ms_per_step_min"] <= rk["ms_per_step_max"] <= d["ms_per_step"] * 1.001
    assert all(np.isfinite(d["config"]["losses"])) and "cpu_baseline" not in d and "roofline" not in d
    names = [b["bucket"] for b in d["comm"]["buckets"]]
    assert names == ["bbox_head.", "neck.", "backbone.layer4.2.", "backbone.layer4.1.", "backbone.layer4.0.", "backbone.layer3.",
                     "backbone.layer2."]
    # the schema the SCALE record is read with (round 6): per-rank step and host-enqueue times, the core set each rank bound
    # itself to (disjoint, before its first GPU call), the traced exchange per bucket, the exposed part of it
    assert len(rk["per_rank_host_enqueue_ms_per_step"]) == 2 and all(0 < v < 1e3 for v in rk["per_rank_host_enqueue_ms_per_step"])
    cores = rk["per_rank_cores"]
    assert len(cores) == 2
    if cores[0] is not None:                              # (None: fewer than two cores allowed, or RADET_BENCH_AFFINITY=0)
        lo = [tuple(int(x) for x in c.split("-")) for c in cores]
        assert lo[0][1] < lo[1][0] or lo[1][1] < lo[0][0], cores
    for key in ("steps", "forward_loss_ms", "backward_ms", "exposed_comm_ms", "bf16_buckets"):
        assert key in d["comm"], key
    for b in d["comm"]["buckets"]:
        assert b["mbytes"] > 0 and 0 <= b["ready_ms"] <= b["done_by_ms"] + 1e-6
    assert d["config"]["step_algorithmic_tflops"] > 0 and "step_frac_of_fp32_mfma_peak" not in d["config"]


@pytest.mark.timeout(600)
def test_bench_script_fails_fast_when_a_rank_is_missing():
    """A rank whose process group cannot form (here: WORLD_SIZE = 2 with only rank 0 started, 6-second limit) leaves with one
    line that names the rank and the phase and a non-zero exit code -- it does not hang until the caller's timeout."""
    import json
    import subprocess
    import sys
    import time
    env = dict(os.environ, RADET_BENCH_SHARE_GPU="1", RADET_BENCH_BACKEND="gloo", RADET_BENCH_COMM_TIMEOUT="6",
               WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=500, env=env, cwd=REPO)
    assert r.returncode in (3, 4) and time.time() - t0 < 300, (r.returncode, r.stdout[-300:], r.stderr[-600:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and "rank 0" in json.loads(lines[0])["error"] and "init_process_group" in json.loads(lines[0])["error"]
