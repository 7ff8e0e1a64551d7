#!/usr/bin/env python
"""Headline benchmark: images/sec of one RADet train step (forward + loss + backward + gradient
all-reduce + grad-clip + AdamW), r50_ycbv_pbr, 640x480, batch 4 per GPU, fp32, synthetic data.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").  The timed region contains only the
product path (HIP kernels through the C ABI + RCCL); the oracle is used solely for the
`cpu_baseline` leg (rank 0, N = 1), as the thing that is timed next to the GPU number.
"""
import argparse
import json
import os
import sys
import time

# main + side (wgrad / slab reduction) + RCCL streams must not share a hardware queue (HIP default: 4 queues, and a
# queue shared by two busy streams serialises them: -12 % on the data-parallel path); read at HIP initialisation
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

IMG_H, IMG_W, PER_GPU_BATCH = 480, 640, 4
TRAIN_FLOP_PER_IMG = 341.1e9      # conv MACs fwd + dgrad + wgrad (frozen stem/layer1), x2
FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense


def synth_objects(rng, G, H=IMG_H, W=IMG_W):
    """Synthetic boxes + visible masks (ellipse in the box, odd objects half occluded) -- SURVEY.md §8d."""
    boxes = np.zeros((G, 4), np.float32)
    masks = np.zeros((G, H, W), np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    for g in range(G):
        w, h = rng.randint(30, 300), rng.randint(30, 300)
        x, y = rng.randint(0, W - w), rng.randint(0, H - h)
        boxes[g] = (x, y, x + w, y + h)
        cx, cy = x + w / 2, y + h / 2
        m = (((xx - cx) / (w / 2)) ** 2 + ((yy - cy) / (h / 2)) ** 2 <= 1)
        if g % 2:
            m[:, :int(cx)] = False
        masks[g] = m
    return boxes, rng.randint(0, 21, G).astype(np.int64), masks


def make_batch(rank, B, device):
    from radet_amd.datasets import LabelAssignment
    rng = np.random.RandomState(1000 + rank)
    g = torch.Generator().manual_seed(1000 + rank)
    img = torch.randn(B, 3, IMG_H, IMG_W, generator=g).to(device)
    boxes, labels, masks, rngs = [], [], [], []
    for i in range(B):
        b, l, m = synth_objects(rng, int(rng.randint(1, 9)))
        boxes.append(b); labels.append(l); masks.append(m)
        rngs.append(np.random.RandomState(123 + rank * B + i))
    la = LabelAssignment(neg_threshold=0.2, positive_num=10, adapt_positive_num=False, balance_sample=True)
    p2g, pw = la.assign_batch(boxes, masks, (IMG_H, IMG_W, 3), rngs=rngs, device=device)   # GPU assigner (product path)
    return img, boxes, labels, p2g, pw


def effective_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota, at most 32."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(steps=3):
    """The oracle (CPU restatement pinned to the reference) timed on this host on the SAME work as one GPU step:
    forward + loss + backward + global-norm clip + AdamW at the per-GPU batch (B = 4, 640x480)."""
    from oracle import assigner as oa, model as om, synth
    nthreads = effective_cores()
    torch.set_num_threads(nthreads)
    det = om.OracleDetector(50, seed=0)
    params = [t for t in det.sd.values() if t.requires_grad]
    opt = torch.optim.AdamW(params, lr=4e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    B = PER_GPU_BATCH
    img = synth.synth_images(0, B)
    gt_b, gt_l, p2g, pw = [], [], [], []
    for i in range(B):
        b, l, m = synth.synth_objects(i, 6)
        a, w = oa.assign_points(b, l, m, (IMG_H, IMG_W, 3), rng=np.random.RandomState(i))
        gt_b.append(torch.from_numpy(b)); gt_l.append(torch.from_numpy(l))
        p2g.append(torch.from_numpy(a)); pw.append(torch.from_numpy(w))
    best, n_timed, t_start = None, 0, time.perf_counter()
    for it in range(steps + 1):
        opt.zero_grad(set_to_none=True)
        t0 = time.perf_counter()
        losses = det.forward_train(img, gt_b, gt_l, p2g, pw)
        om.parse_losses(losses).backward()
        torch.nn.utils.clip_grad_norm_(params, 35.0)
        opt.step()
        dt = time.perf_counter() - t0
        if it > 0 or dt > 15.0:           # a very slow host: keep the (cold) first step as the only sample
            best = dt if best is None else min(best, dt)
            n_timed += 1
        if time.perf_counter() - t_start > 30.0:   # bounded sample (~10-30 s of CPU work)
            break
    return dict(value=round(B / best, 3), unit="images/sec", cores=nthreads, kind="port", cpu=cpu_model(),
                sample=f"oracle (PyTorch-CPU fp32 restatement) forward+loss+backward+clip+AdamW, B={B} 640x480 (one GPU "
                       f"step's work), best of {n_timed} timed step(s), {nthreads} threads")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-mfma-line", action="store_true", help="skip the native-f32-MFMA comparison measurement")
    ap.add_argument("--math", choices=("fp32", "fp32-mfma", "bf16", "bf16-storage"), default="fp32",
                    help="fp32 = the headline metric (BASELINE configs[1]); bf16 = configs[2] arithmetic: conv operands "
                         "rounded to bf16 into the matrix cores, fp32 accumulate / storage / optimizer (secondary line)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("RADET_FORCE_REDUCER") == "1":   # the latter: 1-rank RCCL run of the bucketed exchange
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from radet_amd.models import build_detector
    from radet_amd.utils import Config
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "bop", "r50_ycbv_pbr.py"))
    cfg.model["pretrained"] = None
    torch.manual_seed(0)                      # identical replicas on every rank
    det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).to(device).train()
    rt = det.runtime(math=args.math)
    o = cfg.optimizer
    rt.init_optimizer(lr=o.lr, betas=tuple(o.betas), eps=o.eps, weight_decay=o.weight_decay,
                      max_norm=float(cfg.optimizer_config.grad_clip.max_norm))
    rt.set_loss_from_head(det.bbox_head)
    B = PER_GPU_BATCH
    img, boxes, labels, p2g, pw = make_batch(rank, B, device)
    tg = rt.pack_targets([torch.from_numpy(b) for b in boxes], [torch.from_numpy(l) for l in labels],
                         list(p2g), list(pw))

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    first = None                              # loss triple of optimisation step 1 (random-init weights, seed 0): the
    for _ in range(args.warmup):              # correctness anchor of the run -- equal across runs, boxes and versions
        out_l = rt.train_step(img, tg)
        if first is None:
            first = out_l.clone()
    sync()
    events = None if args.no_kernel_events else []
    rt.engine.tower_events = events
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out_l = rt.train_step(img, tg)
        if first is None:
            first = out_l.clone()             # device-side copy: no host synchronisation inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    rt.engine.tower_events = None
    losses = rt.engine.losses.cpu().numpy()
    assert np.isfinite(losses).all(), f"non-finite losses {losses}"
    t = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        out = {
            "metric": "images/sec train-step, r50_ycbv_pbr 640x480 bs=4/GPU",
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "fp32-mfma": "f32", "bf16": "bf16 operands, f32 accumulate (f32 tensors in HBM)",
                      "bf16-storage": "bf16 tensors + operands, f32 accumulate / loss / optimizer"}[args.math], "data": "synthetic",
            "config": {"workload": "r50_ycbv_pbr bs=4 fp32 forward+loss+backward+allreduce+clip+AdamW (BASELINE configs[1])",
                       "global_batch": world * B, "per_gpu_batch": B, "image": f"{IMG_W}x{IMG_H}",
                       "parallelism": f"dp{world}", "losses_step1": [float(x) for x in first.cpu()],
                       "losses": [float(x) for x in losses],
                       "step_tflops": round(value * TRAIN_FLOP_PER_IMG / 1e12, 2),
                       "step_frac_of_fp32_mfma_peak": round(value / world * TRAIN_FLOP_PER_IMG / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)},
        }
        x3 = bool(rt.engine.x3)
        if args.math.startswith("fp32"):
            out["config"]["arithmetic"] = (
                "f32 tensors, f32-accurate conv GEMMs: every f32 operand is split exactly into three bf16 planes in "
                "registers, 6 of the 9 plane products go through v_mfma_f32_32x32x16_bf16 with f32 accumulation "
                "(measured error vs f64 <= that of v_mfma_f32_32x32x2_f32: tools/x3_probe.py, DESIGN.md 6); "
                "`--math fp32-mfma` / RADET_X3=0 = native f32 MFMA") if x3 else "f32 tensors, v_mfma_f32_32x32x2_f32"
        else:
            out["metric"] += " [bf16 math mode: NOT the headline fp32 metric]"
            out["config"]["workload"] = out["config"]["workload"].replace("fp32", "bf16-math").replace("configs[1]", "configs[2] arithmetic")
        if events and args.math.startswith("fp32"):
            traffic = None          # HBM bytes/launch of the roofline kernel from the committed PMC pass (offline)
            tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json" if x3 else "roofline_traffic_fp32_mfma.json")
            if os.path.exists(tpath):
                with open(tpath) as f:
                    traffic = json.load(f).get("traffic_bytes_per_launch")
            ms_list = [s.elapsed_time(e) for s, e in events]
            avg_ms = float(np.mean(ms_list))
            # pair mode: most tower launches are grouped (cls + reg layer in one launch = 2x the flops)
            flops = rt.engine.tower_gemm_flops()
            pair = rt.engine.tower_mode == "pair"
            hybrid = rt.engine.tower_mode in ("hybrid", "pairbwd")
            if hybrid:
                flops = flops * 2.0     # every tagged launch = cls_convs[i] + reg_convs[i] grouped, forward only
            if pair:
                n_per_step = len(ms_list) // args.steps
                # per step: 4 fwd pairs + 3 dgrad pairs (2 GEMMs each) + 2 single dgrads into dL/dP
                flops = flops * 16.0 / n_per_step
            ach = flops / (avg_ms * 1e-3) / 1e12
            # x3: the kernel runs on the bf16 matrix pipe and issues 6 bf16 MACs per algorithmic f32 MAC: priced against
            # the dense bf16 MFMA peak with the flops it actually issues; the f32-equivalent rate is given next to it
            issued, peak = (6.0 * ach, BF16_MFMA_PEAK_TFLOPS) if x3 else (ach, FP32_MFMA_PEAK_TFLOPS)
            out["roofline"] = {"bound": "mfma", "achieved": round(issued, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(issued / peak, 4), "traffic": traffic,
                               "algorithmic_tflops": round(ach, 2), "frac_of_fp32_mfma_peak": round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                               "kernel": ("conv_igemmg_kernel<128,128,2,2,TAG=9,BK=32,NSTG=2> (f32 operands split into 3 bf16 planes, "
                                          "6 v_mfma_f32_32x32x16_bf16 per K=16 step; `achieved` = issued bf16 flop = 6 x algorithmic)"
                                          if x3 else "conv_igemmg_kernel<128,64,2,2,TAG=1,BK=32,NSTG=3>")
                                         + ": head-tower 3x3 conv GEMM (M=B*6400, N=256, K=2304)"
                                         + ("; forward launches, cls_convs[i] + reg_convs[i] grouped per launch (2 GEMMs), alone on the device"
                                            if hybrid else "; fwd+dgrad launches, cls+reg layers grouped per launch, flop_per_launch = average"
                                            if pair else "; fwd+dgrad launches, cls and reg towers run concurrently on two streams"),
                               "launches": len(ms_list), "avg_us": round(avg_ms * 1e3, 2),
                               "flop_per_launch": flops}
        if world == 1 and x3 and not args.no_mfma_line:
            # the same step with the native f32 matrix instruction: a child process (its own streams and hardware
            # queues; measured inside this process after the main run it shared queues with the first runtime)
            import subprocess
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--math", "fp32-mfma", "--steps", str(args.steps),
                                    "--warmup", str(args.warmup), "--no-cpu-baseline", "--no-kernel-events"],
                                   capture_output=True, text=True, timeout=600)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
                d2 = json.loads(line)
                out["fp32_mfma_native"] = {"value": d2["value"], "unit": "images/sec", "ms_per_step": d2["ms_per_step"],
                                           "losses_step1": d2["config"]["losses_step1"],
                                           "note": "same step, `python bench.py --math fp32-mfma` (v_mfma_f32_32x32x2_f32), child process"}
            except Exception as e:      # the headline line must not depend on the comparison run
                out["fp32_mfma_native"] = {"error": repr(e)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
