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
BF16_MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16 / _f16, dense
# issued 16-bit MACs per algorithmic fp32 MAC of the default fp32 arithmetic (3: fp16 hi / lo planes; 6 with RADET_X3=bf16)
ISSUED_PER_MAC = 6.0 if os.environ.get("RADET_X3", "h2") in ("bf16", "b3") else 3.0


INFER_FLOP_PER_IMG = 120.96e9      # forward conv MACs x 2 of one 640x480 image (backbone + FPN + head)


class ClockSampler:
    """Shader clock and socket power of the visible GPU, read with `rocm-smi --showclocks --showpower` (~80 ms per call) in a
    background thread while a region runs; medians of the samples taken under load.  Only ever started AFTER the timed
    region (every sample starts a Python interpreter, which competes with the launch loop for the host's cores), on rank 0
    of single-GPU runs."""

    def __init__(self):
        import threading
        self.rows, self._stop = [], threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import re
        import shutil
        import subprocess
        # rocm-smi is a `#!/usr/bin/env python3` script: started through the interpreter directly (one exec in the child, no
        # `env` hop), and not at all under rocprofv3 -- its preloaded library initialises the GPU in every process it is
        # inherited by, and a process that has done so must not exec again on this pool
        smi = shutil.which("rocm-smi")
        if smi is None or any(k.startswith("ROCPROF") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
            return
        cmd = [sys.executable, os.path.realpath(smi), "--showclocks", "--showpower"]
        while not self._stop.is_set():
            try:
                txt = subprocess.run(cmd, capture_output=True, text=True, timeout=5).stdout
            except Exception:
                return
            m = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)
            p = re.search(r"Power \(W\): ([0-9.]+)", txt)
            if m and p:
                self.rows.append((time.perf_counter(), int(m.group(1)), float(p.group(1))))

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join(timeout=6)

    def report(self, t0, t1, what):
        rows = [r for r in self.rows if t0 <= r[0] <= t1]
        if not rows:
            return {"samples": 0, "note": "no rocm-smi sample inside the region (not polled under rocprofv3)"}
        clk, pw = sorted(r[1] for r in rows), sorted(r[2] for r in rows)
        return {"sclk_mhz_median": clk[len(clk) // 2], "sclk_mhz_min": clk[0], "sclk_mhz_max": clk[-1],
                "socket_power_w_median": pw[len(pw) // 2], "socket_power_w_max": pw[-1], "samples": len(rows),
                "note": f"rocm-smi --showclocks --showpower polled from a thread during {what} (the interface lags by ~0.2 s: the "
                        "`sustained` window is the steady state); peak sclk 2400 MHz"}


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


def pin_rank_to_cores(local_rank, world):
    """Bind this process to the local_rank-th of `world` equal shares of the cores it is allowed to run on (physical
    neighbours: consecutive ids).  Returns the sorted core list, or None when there is nothing to divide / it is switched off."""
    if os.environ.get("RADET_BENCH_AFFINITY", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        allowed = sorted(os.sched_getaffinity(0))
        per = len(allowed) // world
        if per < 1:
            return None
        mine = allowed[local_rank * per:(local_rank + 1) * per]
        os.sched_setaffinity(0, mine)
        return mine
    except OSError:
        return None


class CommGuard:
    """Watchdog around the start-up collectives of a multi-GPU run: if the armed phase does not finish within `limit`
    seconds the process prints one line saying which phase of which rank hung and exits with code 4 (a blocked RCCL call
    cannot be interrupted from Python)."""

    def __init__(self, rank, limit):
        self.rank, self.limit, self.what, self._t = rank, limit, "", None

    def arm(self, what):
        import threading
        self.disarm()
        self.what = what

        def fire():
            msg = f"rank {self.rank}: {what} did not complete within {self.limit:.0f} s"
            print(json.dumps({"error": msg}), flush=True)
            sys.stderr.write("bench.py: " + msg + "\n")
            sys.stderr.flush()
            os._exit(4)
        self._t = threading.Timer(self.limit, fire)
        self._t.daemon = True
        self._t.start()

    def disarm(self):
        if self._t is not None:
            self._t.cancel()
            self._t = None


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
    """The oracle (CPU restatement pinned to the reference) timed on this host on the same conv work as one GPU step:
    forward + loss + backward + global-norm clip + AdamW at the per-GPU batch (B = 4, 640x480; its own seeded batch of 6
    objects per image -- the GPU step's has 1-8 -- which only moves the loss kernels' share)."""
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
                sample=f"oracle (PyTorch-CPU fp32 restatement) forward+loss+backward+clip+AdamW, B={B} 640x480 (the conv work "
                       f"of one GPU step; 6 objects per image), best of {n_timed} timed step(s), {nthreads} threads")


def kernel_report(events, steps, rt, x3, dt_ev, ms_clean, value, math="fp32"):
    """`roofline` (+ companions) from the per-launch HIP events of the instrumented pass.
    Per kernel instantiation: launches per step, average duration in the step, algorithmic flops, share of the summed
    conv-GEMM time.  `roofline` = the instantiation with the LARGEST share, in the step and alone on the device (every
    distinct launch of it replayed by itself); `roofline_all_conv_gemms` = all of them together; `roofline_tower_forward` =
    the grouped head-tower forward launch (the kernel the previous rounds reported).  The plane arithmetics issue 3 f16 (default)
    or 6 bf16 MACs per algorithmic fp32 MAC: `achieved` is priced in issued 16-bit flop against the dense f16 / bf16 MFMA peak, the
    fp32-equivalent rate is next to it."""
    fam = {}
    for ev in events:
        # one row per kernel SYMBOL, as rocprofv3 counts them (a strided dgrad's class launch carries a suffix in its key:
        # the same symbol -- round 5's record had 55 launches of the dominant symbol here against 58 under the profiler)
        f = fam.setdefault(ev["key"].split(" [")[0], dict(n=0, ms=0.0, flops=0.0, bytes=0.0, evs=[]))
        d = ev["start"].elapsed_time(ev["end"])
        f["n"] += 1; f["ms"] += d; f["flops"] += ev["flops"]; f["bytes"] += ev["bytes"]; f["evs"].append(ev)
    tot_ms = sum(f["ms"] for f in fam.values())
    tot_fl = sum(f["flops"] for f in fam.values())
    def igemm_tag(k):                       # TAG template argument of a conv_igemmg_kernel<BM, BN, WM, WN, TAG, BK, NSTG, SK> key
        if not k.startswith("conv_igemmg_kernel<") or "heuristic" in k:
            return None
        return int(k.split("<")[1].split(">")[0].split(", ")[4])

    def planes(k):                          # issued 16-bit MACs per algorithmic fp32 MAC: 3 (two fp16 planes per operand), 6 (three
        t = igemm_tag(k)                    # bf16 planes), or 0 (native fp32 MFMA)
        if (t is not None and (t & 64)) or ("wgrad" in k and ("fp16" in k or "9q" in k)):
            return 3.0
        if (t is not None and (t & 24) != 0) or "pred3x3" in k or ("wgrad" in k and ("planes" in k or "9p" in k)):
            return 6.0
        return 0.0

    def entry(k, f, alone=False):
        mult, peak = (planes(k), BF16_MFMA_PEAK_TFLOPS) if planes(k) else (1.0, FP32_MFMA_PEAK_TFLOPS)
        if math in ("bf16", "bf16-storage") and not k.startswith("stem"):    # one issued bf16 MAC per algorithmic MAC
            mult, peak = 1.0, BF16_MFMA_PEAK_TFLOPS
        tf = f["flops"] / (f["ms"] * 1e-3) / 1e12
        e = {"kernel": k, "launches_per_step": round(f["n"] / steps, 2), "avg_us": round(f["ms"] / f["n"] * 1e3, 2),
             "flop_per_launch": f["flops"] / f["n"], "algorithmic_bytes_per_launch": round(f["bytes"] / f["n"]),
             "share_of_conv_gemm_time": round(f["ms"] / tot_ms, 4),
             "algorithmic_tflops": round(tf, 2), "achieved": round(tf * mult, 2), "peak": peak, "unit": "TFLOP/s",
             "frac": round(tf * mult / peak, 4), "bound": "mfma"}
        if alone:                           # every distinct launch of one step replayed alone (3 warm + 5 timed, back to back)
            per_step = f["evs"][:max(1, f["n"] // steps)]
            a_ms = 0.0
            for ev in per_step:
                for _ in range(3):
                    ev["replay"]()
                torch.cuda.synchronize()
                s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_.record()
                for _ in range(5):
                    ev["replay"]()
                e_.record(); e_.synchronize()
                a_ms += s_.elapsed_time(e_) / 5
            a_tf = sum(ev["flops"] for ev in per_step) / (a_ms * 1e-3) / 1e12
            e["alone"] = {"avg_us": round(a_ms / len(per_step) * 1e3, 2), "algorithmic_tflops": round(a_tf, 2),
                          "achieved": round(a_tf * mult, 2), "frac": round(a_tf * mult / peak, 4),
                          "note": "the same launches one at a time on an otherwise idle device"}
        return e
    order = sorted(fam.items(), key=lambda kv: -kv[1]["ms"])
    dom_k, dom = order[0]
    roof = entry(dom_k, dom, alone=True)
    traffic, tnote = None, None             # HBM bytes / launch from the committed PMC passes, if they cover this kernel
    tname = "roofline_traffic_bf16_storage.json" if math == "bf16-storage" else "roofline_traffic.json"
    tpath = os.path.join(ROOT, "profiles", tname)
    tj = {}
    if os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)

    def pmc_traffic(k):                      # (the same PMC passes cover every kernel of the step: "all_kernels")
        if k.startswith("conv_wgradg_kernel<") and "fp16 hi/lo in registers" in k:
            # the launchers' readable key -> the symbol rocprofv3 reports: <BM, BN, WM, WN, 3 (= fp16 hi / lo), pixels per stage, pixel groups>
            bm, bn = k.split("<")[1].split(">")[0].split(", ")
            px, pg = (64, 4) if "64 px / 4 waves" in k else ((32, 2) if "32 px / 2 waves" in k else ((32, 1) if ", 32 px" in k else (16, 1)))
            wm, wn = (1, 4) if bm == "32" else (2, 2)
            k = f"conv_wgradg_kernel<{bm}, {bn}, {wm}, {wn}, 3, {px}, {pg}>"
        kk = k.split(" (")[0].replace(" ", "")
        for name, v in tj.get("all_kernels", {}).items():
            nn = name.replace(" ", "")
            if nn == kk or nn.startswith(kk + "<") or (kk.endswith(">") and nn.startswith(kk)):
                return v
        return None
    if tj.get("kernel", "").replace(" ", "") == dom_k.replace(" ", ""):
        traffic, tnote = tj.get("traffic_bytes_per_launch"), tj.get("note")
    elif pmc_traffic(dom_k) is not None:
        traffic, tnote = pmc_traffic(dom_k), "average over this kernel's launches of one step (several conv shapes share the symbol)"
    roof["traffic"] = traffic
    if traffic and roof.get("algorithmic_bytes_per_launch"):
        roof["traffic_over_algorithmic"] = round(traffic / roof["algorithmic_bytes_per_launch"], 2)
    roof["traffic_source"] = (f"profiles/{tname} (committed rocprofv3 PMC pass of this kernel: FETCH_SIZE / WRITE_SIZE in separate "
                              "runs, tools/pmc_traffic.py); NOT measured in this run") if traffic is not None else None
    if tnote:
        roof["traffic_note"] = tnote
    roof["selection"] = ("largest share of the summed conv-GEMM kernel time of the step (HIP events around every conv launch, "
                         "second pass of the same K steps)")
    roof["note"] = ("`frac` is priced on the launch's duration IN the step, where launches on the engine's four streams share the "
                    "CUs (the tower weight gradients run next to the two dgrad chains by design: a workgroup of either owns a CU and "
                    "their tiles interleave), so a launch's duration covers other launches' work too; `alone` = the same launches "
                    "one at a time.  `peak` is the 2.4 GHz figure; `clock_power` has the clock the step sustains (the fp16 hi / lo "
                    "arithmetic stays below the socket power limit, the bf16-triple scheme of rounds 2-4 ran at it).")
    def with_traffic(e, t):                 # HBM / fabric bytes per launch from the committed PMC passes (profiles/<tname>), if covered
        e["traffic"] = t
        e["traffic_over_algorithmic"] = round(t / e["algorithmic_bytes_per_launch"], 2) if t and e["algorithmic_bytes_per_launch"] else None
        return e
    mult_all = (3.0 if getattr(rt.engine, "h2", False) else 6.0) if x3 else 1.0
    peak_all = BF16_MFMA_PEAK_TFLOPS if x3 else FP32_MFMA_PEAK_TFLOPS
    if math in ("bf16", "bf16-storage"):
        mult_all, peak_all = 1.0, BF16_MFMA_PEAK_TFLOPS
    all_tf = tot_fl / (tot_ms * 1e-3) / 1e12
    rep = {"roofline": roof,
           "roofline_all_conv_gemms": {
               "bound": "mfma", "flop_per_step": tot_fl / steps, "summed_kernel_ms_per_step": round(tot_ms / steps, 3),
               "algorithmic_tflops_over_summed_kernel_time": round(all_tf, 2),
               "achieved": round(all_tf * mult_all, 2), "peak": peak_all, "unit": "TFLOP/s", "frac": round(all_tf * mult_all / peak_all, 4),
               "step_level": {"algorithmic_tflops": round(value * TRAIN_FLOP_PER_IMG / 1e12, 2),
                              "achieved": round(value * TRAIN_FLOP_PER_IMG / 1e12 * mult_all, 2),
                              "frac": round(value * TRAIN_FLOP_PER_IMG / 1e12 * mult_all / peak_all, 4),
                              "note": "images/s x 341.1 GFLOP/img (SURVEY 8d) over the wall time of the step: streams overlap, so this is above the summed-kernel-time figure"},
               "kernels": [with_traffic(entry(k, f), pmc_traffic(k)) for k, f in order[:10]]},
           "kernel_events": {"ms_per_step_with_events": round(dt_ev / steps * 1e3, 3), "ms_per_step": round(ms_clean, 3),
                             "conv_launches_per_step": round(len(events) / steps, 1)}}
    rep["stages"] = stage_table(events, steps, mult_all, peak_all)
    tg = rt.engine.tower_gemm_flops()
    tower = [(k, f) for k, f in order if igemm_tag(k) is not None and (igemm_tag(k) & 1) and     # the tagged symbol: forward launches only
             min(abs(f["flops"] / f["n"] - tg), abs(f["flops"] / f["n"] - 2.0 * tg)) < 1.0]
    if tower:
        k, f = tower[0]
        pair = abs(f["flops"] / f["n"] - 2.0 * tg) < 1.0
        rep["roofline_tower_forward"] = entry(k, f, alone=not pair)
        rep["roofline_tower_forward"]["note"] = (
            "grouped cls_convs[i] + reg_convs[i] forward launch (2 GEMMs of M = B*6400, N = 256, K = 2304), alone on the device in the step"
            if pair else
            "one head-tower forward GEMM (cls_convs[i] or reg_convs[i]: M = B*6400, N = 256, K = 2304).  The two towers are two chains "
            "on two streams, so in the step two of these launches (and the other chain's GroupNorm) share the device: the in-step "
            "duration of ONE launch covers up to two launches' work; `alone` = the same launch by itself")
    return rep


def stage_table(events, steps, mult, peak):
    """Per part of the detector (stem, layer1..4, neck, head) and kind of launch (fwd / dgrad / wgrad): launches per step, summed
    kernel time, WALL time from the first launch's start event to the last launch's end event (launches of one kind and stage
    overlap with other streams' work, and wgrads run next to the dgrad chain, so walls of different rows overlap), algorithmic
    flop, and the rate over the wall time as a fraction of the pipe the step's arithmetic runs on."""
    per = len(events) // steps
    acc = {}
    for st in range(steps):
        chunk = events[st * per:(st + 1) * per]
        ref = chunk[0]["start"]
        spans = {}
        for ev in chunk:
            k = (ev.get("stage") or "?", ev.get("kind") or "fwd")
            t0, t1 = ref.elapsed_time(ev["start"]), ref.elapsed_time(ev["end"])
            sp = spans.setdefault(k, [t0, t1, 0.0, 0.0, 0])
            sp[0], sp[1] = min(sp[0], t0), max(sp[1], t1)
            sp[2] += t1 - t0; sp[3] += ev["flops"]; sp[4] += 1
        for k, sp in spans.items():
            a = acc.setdefault(k, [0.0, 0.0, 0.0, 0, 0.0])
            a[0] += sp[1] - sp[0]; a[1] += sp[2]; a[2] += sp[3]; a[3] += sp[4]; a[4] += sp[0]
    order = ["stem", "layer1", "layer2", "layer3", "layer4", "neck", "head"]
    rows = []
    for (stage, kind), a in sorted(acc.items(), key=lambda kv: (order.index(kv[0][0]) if kv[0][0] in order else 99,
                                                                 ("fwd", "dgrad", "wgrad").index(kv[0][1]))):
        wall_ms, sum_ms, fl = a[0] / steps, a[1] / steps, a[2] / steps
        tf = fl / (wall_ms * 1e-3) / 1e12 if wall_ms > 0 else 0.0
        rows.append({"stage": stage, "kind": kind, "launches_per_step": round(a[3] / steps, 1), "wall_us": round(wall_ms * 1e3, 1),
                     "summed_kernel_us": round(sum_ms * 1e3, 1), "starts_at_us": round(a[4] / steps * 1e3, 1),
                     "gflop": round(fl / 1e9, 2), "algorithmic_tflops_over_wall": round(tf, 1),
                     "frac_of_pipe_over_wall": round(tf * (1.0 if stage == "stem" else mult) / (FP32_MFMA_PEAK_TFLOPS if stage == "stem" else peak), 4)})
    return {"rows": rows,
            "note": "from the HIP events of the instrumented pass (conv GEMM launches only: GroupNorm / loss / pooling / optimizer "
                    "kernels are not in it); wall = first start .. last end of the row's launches inside one step, averaged over "
                    "the steps; the stem runs on the fp32 MFMA pipe (priced against 157.3 TFLOP/s), everything else against "
                    "the pipe of the step's arithmetic (x3 issued f16 MACs / 2500 for the default fp32 mode; x6 bf16 MACs with "
                    "RADET_X3=bf16)"}


def _child(argv, timeout=900):
    import subprocess
    r = subprocess.run([sys.executable] + argv, capture_output=True, text=True, timeout=timeout)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if not line:
        raise RuntimeError((r.stderr or r.stdout)[-300:])
    return json.loads(line[-1])


def extras(args):
    """Secondary measurements, each in a child process (its own streams / hardware queues): BASELINE configs[3] (inference)
    and configs[4] (R101 800x800 bs 2) as short runs, and the headline step on seeded trained-like parameters."""
    out = {}
    cfgs = os.path.join(ROOT, "tools", "bench_configs.py")
    for key, argv in (("infer", [cfgs, "infer", "--json", "--images", "1000"]), ("r101", [cfgs, "r101", "--json", "--steps", "8"])):
        try:
            out[key] = _child(argv)
        except Exception as e:                   # the headline line must not depend on the secondary runs
            out[key] = {"error": repr(e)[:200]}
    me = os.path.abspath(__file__)
    short = ["--steps", str(args.steps), "--warmup", str(args.warmup), "--no-cpu-baseline", "--no-mfma-line", "--no-extras"]
    try:                                         # BASELINE configs[2] arithmetic on one GPU (bf16 tensors + operands)
        d = _child([me, "--math", "bf16-storage"] + short)
        out["bf16_storage"] = {
            "value": d["value"], "unit": "images/sec", "ms_per_step": d["ms_per_step"], "dtype": d["dtype"],
            "host_enqueue_ms_per_step": d.get("host_enqueue_ms_per_step"), "losses_step1": d["config"]["losses_step1"],
            "step_level": {"achieved": round(d["value"] * TRAIN_FLOP_PER_IMG / 1e12, 2), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(d["value"] * TRAIN_FLOP_PER_IMG / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4)},
            "roofline": {k: d["roofline"][k] for k in ("kernel", "avg_us", "achieved", "peak", "frac", "share_of_conv_gemm_time")
                         if k in d.get("roofline", {})},
            "clock_power": d.get("clock_power"),
            "note": "same step, `python bench.py --math bf16-storage` (BASELINE configs[2] arithmetic on ONE GPU: bf16 activations / "
                    "folded weights / activation gradients in HBM, v_mfma_f32_32x32x16_bf16 with f32 accumulation, f32 loss / "
                    "GroupNorm statistics / master weights / AdamW); one issued bf16 MAC per algorithmic MAC"}
    except Exception as e:
        out["bf16_storage"] = {"error": repr(e)[:200]}
    try:                                         # the reference's own operating point: samples_per_gpu = 16
        d = _child([me, "--batch", "16", "--no-kernel-events"] + short)
        out["bs16"] = {
            "value": d["value"], "unit": "images/sec", "ms_per_step": d["ms_per_step"], "per_gpu_batch": 16,
            "host_enqueue_ms_per_step": d.get("host_enqueue_ms_per_step"),
            "step_level": {"achieved": round(d["value"] * TRAIN_FLOP_PER_IMG / 1e12 * ISSUED_PER_MAC, 2), "peak": BF16_MFMA_PEAK_TFLOPS,
                           "unit": "TFLOP/s", "frac": round(d["value"] * TRAIN_FLOP_PER_IMG / 1e12 * ISSUED_PER_MAC / BF16_MFMA_PEAK_TFLOPS, 4)},
            "clock_power": d.get("clock_power"),
            "note": "same fp32 step at the reference config's samples_per_gpu = 16 (configs/bop/r50_ycbv_pbr.py:85), "
                    "`python bench.py --batch 16`; secondary -- the headline is BASELINE's bs 4"}
    except Exception as e:
        out["bs16"] = {"error": repr(e)[:200]}
    try:
        d = _child([os.path.abspath(__file__), "--weights", "synth", "--steps", str(args.steps), "--warmup", str(args.warmup),
                    "--no-cpu-baseline", "--no-kernel-events", "--no-mfma-line", "--no-extras"])
        out["synthetic_trained_like_weights"] = {
            "value": d["value"], "unit": "images/sec", "ms_per_step": d["ms_per_step"], "clock_power": d.get("clock_power"),
            "losses_step1": d["config"]["losses_step1"],
            "note": "same step, `python bench.py --weights synth`: seeded trained-like parameters / running statistics "
                    "(radet_amd/utils/synth_init.py) -> dense, decorrelated activations; the plane arithmetic runs against "
                    "the power envelope, so its speed depends on the operands' bit activity (DESIGN.md 6)"}
    except Exception as e:
        out["synthetic_trained_like_weights"] = {"error": repr(e)[:200]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-mfma-line", action="store_true", help="skip the native-f32-MFMA comparison measurement")
    ap.add_argument("--no-clock-sampler", action="store_true", help="skip the rocm-smi clock / power window after the timed region")
    ap.add_argument("--prefetch", action="store_true",
                    help="hand the next batch to train_step: its frozen stem + layer1 then run during this step's backward pass "
                         "(measured neutral on one GPU: the backward pass is throughput-bound, DESIGN.md; off by default)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary measurements (inference config 4, R101 config 5, synthetic trained-like weights)")
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH,
                    help="images per GPU (default 4 = BASELINE configs[1], the headline; 16 = the reference config's samples_per_gpu)")
    ap.add_argument("--weights", choices=("init", "synth"), default="init",
                    help="init = the detector's own random initialisation (headline); synth = seeded trained-like parameters "
                         "and running statistics (radet_amd/utils/synth_init.py): dense, decorrelated activations")
    ap.add_argument("--math", choices=("fp32", "fp32-mfma", "bf16", "bf16-storage"), default="fp32",
                    help="fp32 = the headline metric (BASELINE configs[1]); bf16 = configs[2] arithmetic: conv operands "
                         "rounded to bf16 into the matrix cores, fp32 accumulate / storage / optimizer (secondary line)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` started plainly: become the launcher -- N worker processes through torch.distributed.run
        # (one rank per GPU, rendezvous on 127.0.0.1), started as a CHILD before this process has touched the GPU, rank 0's JSON
        # line relayed.  (What the reference's tools/train.py:117-124 leaves to its launcher + init_dist.)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        results = [ln for ln in lines if '"metric"' in ln]
        for ln in (results or lines[:1]):               # rank 0's result line, else the first rank's one-line reason
            print(ln, flush=True)
        sys.exit(r.returncode if (r.returncode or results) else 5)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook (tests/test_gpu_distributed.py): RADET_BENCH_SHARE_GPU=1 lets the ranks of a 1-GPU box share device 0 and
    # RADET_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device), so that the N > 1 code path of this
    # script runs before the driver's 8-GPU run does
    share = os.environ.get("RADET_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("RADET_BENCH_BACKEND", "nccl")
    if share:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    # Host cores: each rank of a multi-GPU run binds itself to its own share of the cores this job may use, BEFORE the first GPU
    # call (os.sched_setaffinity on the running process: no taskset, no re-exec) -- eight ranks' Python loops and HIP runtime
    # threads on one shared pool migrate and preempt each other; with the launch tape a rank needs ~3 ms of one core per step.
    # RADET_BENCH_AFFINITY=0 leaves the scheduler alone.
    cpus = pin_rank_to_cores(local_rank if not share else rank, world) if world > 1 else None
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("RADET_FORCE_REDUCER") == "1":   # the latter: 1-rank RCCL run of the bucketed exchange
        os.environ.setdefault("TORCH_NCCL_ENABLE_TIMING", "1")      # per-collective durations for the `comm` report
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # a rank whose process group does not come up, or whose first collective does not complete, ends the job with ONE line
        # and a non-zero exit code instead of hanging until the caller's timeout (the first N = 8 run is the driver's, unattended)
        limit = float(os.environ.get("RADET_BENCH_COMM_TIMEOUT", "240"))
        guard = CommGuard(rank, limit)
        try:
            import datetime
            guard.arm("init_process_group")
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device,
                                        timeout=datetime.timedelta(seconds=limit))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=limit))
            guard.arm("first all_reduce")
            probe = torch.ones(1024, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(probe)
            if probe.is_cuda:
                torch.cuda.synchronize()
            if float(probe[0]) != float(world):
                raise RuntimeError(f"first all_reduce returned {float(probe[0])}, expected {world}")
            guard.disarm()
        except Exception as e:               # noqa: BLE001 -- whatever it is, say it in one line and leave
            guard.disarm()
            print(json.dumps({"error": f"rank {rank}/{world}: {guard.what} failed: {type(e).__name__}: {str(e)[:300]}"}), flush=True)
            sys.stderr.write(f"bench.py: rank {rank}/{world}: {guard.what} failed: {type(e).__name__}: {str(e)[:300]}\n")
            os._exit(3)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from radet_amd.models import build_detector
    from radet_amd.utils import Config
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "bop", "r50_ycbv_pbr.py"))
    cfg.model["pretrained"] = None
    torch.manual_seed(0)                      # identical replicas on every rank
    det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).to(device).train()
    if args.weights == "synth":
        from radet_amd.utils.synth_init import synth_fill
        synth_fill(det, seed=0)
    rt = det.runtime(math=args.math)
    o = cfg.optimizer
    rt.init_optimizer(lr=o.lr, betas=tuple(o.betas), eps=o.eps, weight_decay=o.weight_decay,
                      max_norm=float(cfg.optimizer_config.grad_clip.max_norm))
    rt.set_loss_from_head(det.bbox_head)
    B = args.batch
    img, boxes, labels, p2g, pw = make_batch(rank, B, device)
    tg = rt.pack_targets([torch.from_numpy(b) for b in boxes], [torch.from_numpy(l) for l in labels],
                         list(p2g), list(pw))

    # the next step's batch is known one step ahead (a data loader prefetches it): its frozen stem + layer1 run during the
    # current step's backward pass (runtime.train_step, next_img).  Synthetic data: the same resident batch every step, so
    # "next" is the same tensor -- the prefix is nevertheless computed once per step, into the buffer set the step after uses.
    nxt = img if args.prefetch else None

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    first = None                              # loss triple of optimisation step 1 (random-init weights, seed 0): the
    for _ in range(args.warmup):              # correctness anchor of the run -- equal across runs, boxes and versions
        out_l = rt.train_step(img, tg, next_img=nxt)
        if first is None:
            first = out_l.clone()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out_l = rt.train_step(img, tg, next_img=nxt)
        if first is None:
            first = out_l.clone()             # device-side copy: no host synchronisation inside the timed region
    t_enq = time.perf_counter() - t0          # the Python loop body alone: every launch of K steps enqueued, nothing awaited
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    t_end = t0 + dt
    dt_rank = dt
    losses = rt.engine.losses.cpu().numpy()
    # the same K steps once more with a pair of HIP events around EVERY conv GEMM launch, on the stream it is launched on
    # (radet_amd.kernels.EVENTS): the per-kernel table behind `roofline`.  Kept out of the region above so that `value` is
    # the uninstrumented step; the instrumented step time is reported next to it.
    events, dt_ev = None, None
    if rank == 0 and world == 1 and not args.no_kernel_events:
        from radet_amd import kernels as K
        torch.cuda.synchronize()
        K.EVENTS = events = []
        t1 = time.perf_counter()
        for _ in range(args.steps):
            rt.train_step(img, tg, next_img=nxt)
        torch.cuda.synchronize()
        dt_ev = time.perf_counter() - t1
        K.EVENTS = None
    sampler = ClockSampler() if (rank == 0 and world == 1 and not args.no_clock_sampler) else None
    if sampler is not None:
        # clock / power of the steady state: the same step kept running for ~1.5 s (untimed, after everything that is timed)
        # while rocm-smi is polled from a thread
        sampler.__enter__()
        t_keep0 = time.perf_counter()
        while time.perf_counter() - t_keep0 < 1.5:
            for _ in range(10):
                rt.train_step(img, tg, next_img=nxt)
            torch.cuda.synchronize()
        t_keep1 = time.perf_counter()
        sampler.__exit__()
    assert np.isfinite(losses).all(), f"non-finite losses {losses}"
    cdev = device if (not dist.is_initialized() or dist.get_backend() == "nccl") else torch.device("cpu")   # (gloo: host tensors)
    t = torch.tensor([dt], device=cdev, dtype=torch.float64)
    rank_ms = None
    if world > 1:
        per_rank = [torch.zeros(4, device=cdev, dtype=torch.float64) for _ in range(world)]
        mine = [dt_rank, t_enq, float(cpus[0]) if cpus else -1.0, float(len(cpus)) if cpus else 0.0]
        dist.all_gather(per_rank, torch.tensor(mine, device=cdev, dtype=torch.float64))
        rank_ms = [[round(float(v[0]) / args.steps * 1e3, 3), round(float(v[1]) / args.steps * 1e3, 3), int(v[2]), int(v[3])]
                   for v in per_rank]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    comm = None
    if rt.reducer is not None:                # data-parallel runs: a few more steps with the bucket exchange traced
        rt.reducer.enable_trace(True)
        for _ in range(min(args.steps, 8)):
            rt.train_step(img, tg, next_img=nxt)
        comm = rt.comm_report()
        rt.reducer.enable_trace(False)
        if comm is not None:
            comm["note"] = ("this rank; ms after the start of the backward pass: `ready` = bucket handed to RCCL (its slab reduction "
                            "finished on the side stream), `allreduce_ms` = RCCL's own start-to-end time of the collective, `done_by` = "
                            "seen complete by the main stream (upper bound); exposed = what the main stream waits for the exchange "
                            "after its last backward kernel (clip + AdamW start later by this much)")
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        out = {
            "metric": f"images/sec train-step, r50_ycbv_pbr 640x480 bs={B}/GPU",
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "fp32-mfma": "f32", "bf16": "bf16 operands, f32 accumulate (f32 tensors in HBM)",
                      "bf16-storage": "bf16 tensors + operands, f32 accumulate / loss / optimizer"}[args.math], "data": "synthetic",
            "config": {"workload": "r50_ycbv_pbr bs=4 fp32 forward+loss+backward+allreduce+clip+AdamW (BASELINE configs[1])",
                       "global_batch": world * B, "per_gpu_batch": B, "image": f"{IMG_W}x{IMG_H}",
                       "parallelism": f"dp{world}", "losses_step1": [float(x) for x in first.cpu()],
                       "losses": [float(x) for x in losses],
                       "step_algorithmic_tflops": round(value * TRAIN_FLOP_PER_IMG / 1e12, 2),
                       "launch_tape": rt.tape_stats()},
        }
        out["host_enqueue_ms_per_step"] = round(t_enq / args.steps * 1e3, 3)
        out["host"] = {"cores": effective_cores(), "cpu": cpu_model(),
                       "note": "host_enqueue_ms_per_step = wall time of the Python loop body of the timed region (all launches of a "
                               "step enqueued, no synchronisation) per step, this rank; the step is GPU-bound while it stays below "
                               "ms_per_step"}
        if B != PER_GPU_BATCH:
            out["metric"] += " [NOT the headline: BASELINE configs[1] is bs 4]"
            out["config"]["workload"] = out["config"]["workload"].replace("bs=4", f"bs={B}").replace("(BASELINE configs[1])", "(secondary)")
        if rank_ms is not None:
            out["ranks"] = {"ms_per_step_min": min(r[0] for r in rank_ms), "ms_per_step_max": max(r[0] for r in rank_ms),
                            "host_enqueue_ms_per_step_max": max(r[1] for r in rank_ms), "per_rank_ms_per_step": [r[0] for r in rank_ms],
                            "per_rank_host_enqueue_ms_per_step": [r[1] for r in rank_ms],
                            "per_rank_cores": [(f"{r[2]}-{r[2] + r[3] - 1}" if r[3] > 0 else None) for r in rank_ms],
                            "launch_tape": rt.tape_stats(),
                            "note": "each rank's own wall time of the K timed steps (its synchronize + the closing barrier included); a "
                                    "straggler shows as min << max"}
        if sampler is not None:
            out["clock_power"] = sampler.report(t_keep0, t_keep1, "~1.5 s of the same step right after the timed region")
        if comm is not None:
            out["comm"] = comm
        x3 = bool(rt.engine.x3)
        if args.math.startswith("fp32"):
            h2 = bool(getattr(rt.engine, "h2", False))
            out["config"]["arithmetic"] = (
                "f32 tensors, f32-accurate conv GEMMs: every f32 operand is scaled by an exact power of two (from the tensor's "
                "largest magnitude, tracked by the kernel that writes it) and split into two f16 numbers hi + 2^-11 lo (in "
                "registers; the head towers' activations / gradients / weights once, by the kernel that produces them); "
                "hi hi' + 2^-11 (hi lo' + lo hi') goes through three v_mfma_f32_32x32x16_f16 per K = 16 step with f32 accumulation "
                "(error vs f64 <= that of v_mfma_f32_32x32x2_f32 incl. operands spanning 2^24 inside a tensor: "
                "tests/test_gpu_kernels.py::test_fp32_from_fp16_pairs_is_as_accurate_as_the_fp32_mfma, DESIGN.md); "
                "RADET_X3=bf16 = the 6-product bf16-plane scheme of rounds 2-4, `--math fp32-mfma` / RADET_X3=0 = native f32 MFMA"
                if h2 else
                "f32 tensors, f32-accurate conv GEMMs: every f32 operand is split exactly into three bf16 planes (in "
                "registers; the head towers' activations / gradients / weights once, by the kernel that produces them), "
                "6 of the 9 plane products go through v_mfma_f32_32x32x16_bf16 with f32 accumulation "
                "(measured error vs f64 <= that of v_mfma_f32_32x32x2_f32: tools/x3_probe.py, DESIGN.md 6); "
                "`--math fp32-mfma` / RADET_X3=0 = native f32 MFMA") if x3 else "f32 tensors, v_mfma_f32_32x32x2_f32"
        else:
            out["metric"] += " [bf16 math mode: NOT the headline fp32 metric]"
            out["config"]["workload"] = out["config"]["workload"].replace("fp32", "bf16-math").replace("configs[1]", "configs[2] arithmetic")
        if events:
            out.update(kernel_report(events, args.steps, rt, x3, dt_ev, ms, value, args.math))
        if world == 1 and x3 and not args.no_mfma_line:
            # the same step with the native f32 matrix instruction: a child process (its own streams and hardware
            # queues; measured inside this process after the main run it shared queues with the first runtime)
            import subprocess
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--math", "fp32-mfma", "--steps", str(args.steps),
                                    "--warmup", str(args.warmup), "--no-cpu-baseline", "--no-kernel-events", "--no-extras"],
                                   capture_output=True, text=True, timeout=600)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
                d2 = json.loads(line)
                out["fp32_mfma_native"] = {"value": d2["value"], "unit": "images/sec", "ms_per_step": d2["ms_per_step"],
                                           "losses_step1": d2["config"]["losses_step1"],
                                           "note": "same step, `python bench.py --math fp32-mfma` (v_mfma_f32_32x32x2_f32), child process"}
            except Exception as e:      # the headline line must not depend on the comparison run
                out["fp32_mfma_native"] = {"error": repr(e)[:200]}
        if world == 1 and args.math == "fp32" and args.weights == "init" and not args.no_extras:
            ex = extras(args)
            out.update(ex)
            sw = ex.get("synthetic_trained_like_weights", {})
            if isinstance(sw.get("ms_per_step"), (int, float)):
                # next to `value`: the same step on NON-degenerate operands (the reference's init zeroes every Bottleneck's
                # norm3.weight, backbones/resnet.py:600-618, so a third of the headline's bottleneck convs multiply zeros)
                out["value_trained_like_weights"] = sw["value"]
                out["ms_per_step_trained_like_weights"] = sw["ms_per_step"]
            # the secondary numbers once more inside `config` (numbers only): the driver's record keeps `config` verbatim
            sec = {}
            for key, fields in (("infer", ("value",)), ("r101", ("ms_per_step", "value")), ("bf16_storage", ("value", "ms_per_step")),
                                ("bs16", ("value", "ms_per_step")), ("synthetic_trained_like_weights", ("value",))):
                for fld in fields:
                    v = ex.get(key, {}).get(fld)
                    if isinstance(v, (int, float)):
                        sec[f"{key}_{fld}"] = v
            v = ex.get("infer", {}).get("step_level", {}).get("frac")
            if isinstance(v, (int, float)):
                sec["infer_frac"] = v
            if isinstance(out.get("fp32_mfma_native", {}).get("value"), (int, float)):
                sec["fp32_mfma_native_value"] = out["fp32_mfma_native"]["value"]
            sec["host_enqueue_ms"] = out["host_enqueue_ms_per_step"]
            out["config"]["secondary"] = sec
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
