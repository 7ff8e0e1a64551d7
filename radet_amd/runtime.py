"""Per-detector device runtime: flat parameter / gradient / optimiser-state arenas, the HIP Engine,
target packing, the native train step (forward + hand-written backward + RCCL all-reduce overlapped
with backward + fused clip/AdamW) and inference (decode + NMS).

Replaces what the reference obtains from torch autograd + mmcv's DDP wrapper / OptimizerHook
(radet/apis/train.py:73-126, radet/models/detectors/base.py:185-253).
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from . import kernels as K
from .engine import Engine


def _align(n, a=4):
    return (n + a - 1) // a * a


class FlatParams:
    """Re-points every floating-point parameter / buffer of `module` into flat fp32 arenas.

    Trainable parameters (module order: backbone -> neck -> head, which is also the order of the
    engine's conv table) form one arena with a same-layout gradient arena; the rest (frozen
    parameters, BN running statistics) a second one."""

    def __init__(self, module, device):
        named = [(n, p) for n, p in module.named_parameters()]
        train = [(n, p) for n, p in named if p.requires_grad]
        frozen = [(n, p) for n, p in named if not p.requires_grad]
        bufs = [(n, b) for n, b in module.named_buffers() if b.is_floating_point()]
        self.train_names = [n for n, _ in train]
        self.offsets = {}
        off = 0
        for n, p in train:
            if p.numel() >= 4:
                off = _align(off)
            self.offsets[n] = off
            off += p.numel()
        self.n_train = _align(off)
        self.params = torch.zeros(self.n_train, device=device)
        self.grads = torch.zeros(self.n_train, device=device)
        self.p, self.g = {}, {}
        # name -> the tensor objects through which torch code can write this parameter: the arena view and the module's
        # Parameter (same storage, separate version counters); Engine.fold() watches both
        self.watch = {}
        for n, p in train:
            o = self.offsets[n]
            v = self.params[o:o + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
            self.p[n] = v
            self.watch[n] = (v, p)
            self.g[n] = self.grads[o:o + p.numel()].view(p.shape)
        off = 0
        foffs = {}
        for n, t in frozen + bufs:
            off = _align(off)
            foffs[n] = off
            off += t.numel()
        self.frozen = torch.zeros(_align(off), device=device)
        for n, p in frozen:
            v = self.frozen[foffs[n]:foffs[n] + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
            self.p[n] = v
            self.watch[n] = (v, p)
        for n, b in bufs:
            v = self.frozen[foffs[n]:foffs[n] + b.numel()].view(b.shape)
            v.copy_(b)
            mod, _, leaf = n.rpartition(".")
            owner = module.get_submodule(mod) if mod else module
            owner._buffers[leaf] = v
            self.p[n] = v
        self._probe = train[0][1] if train else None

    def group_ranges(self, prefixes):
        """Contiguous [begin, end) ranges of the trainable arena per name-prefix group (in arena order)."""
        out = []
        for pf in prefixes:
            names = [n for n in self.train_names if n.startswith(pf)]
            if not names:
                continue
            b = self.offsets[names[0]]
            last = names[-1]
            e = self.offsets[last] + self.p[last].numel()
            out.append((pf, b, e))
        return out

    def still_bound(self):
        p = self._probe
        return p is None or (p.data_ptr() == self.p[self.train_names[0]].data_ptr())


def compute_buckets(flat, conv_names, conv_trainable, split=("backbone.layer4.",), min_bytes=15e6):
    """Gradient buckets in backward order: (prefix, conv-table range, gradient-arena range).
    Arena ranges are contiguous, disjoint and together cover every trainable parameter.
    A stage named in `split` is cut into one bucket per run of bottleneck blocks of >= min_bytes (fp32), last block first:
    layer4 holds 47 % of all gradient bytes (59.9 MB) and is the first backbone stage of the backward pass, so as ONE message
    its exchange could only start when the whole stage's weight gradients were reduced (3.97 ms into a 5.93-ms backward
    pass); per block (17.8 + 17.8 + 24.3 MB) the first message leaves a third of the way through the stage.  xGMI links
    want few large messages, hence the 15-MB floor (layer3's six blocks of 4.5 MB stay one 28-MB bucket)."""
    def rng(pf):
        idx = [i for i, n in enumerate(conv_names) if n.startswith(pf)]
        return (idx[0], idx[-1] + 1) if idx else None

    def bucket(pf, groups, r, blocks=None):
        arena = flat.group_ranges(groups)
        d = dict(prefix=pf, convs=r, arena=(min(a[1] for a in arena), max(a[2] for a in arena)))
        if blocks is not None:
            d["blocks"] = blocks                  # (stage index, first block, last block) of a per-block bucket
        return d

    buckets = []
    for pf in ("bbox_head.", "neck.", "backbone.layer4.", "backbone.layer3.", "backbone.layer2.", "backbone.layer1.",
               "backbone.conv1"):
        r = rng(pf)
        if r is None or not any(conv_trainable[r[0]:r[1]]):
            continue
        if pf in split and os.environ.get("RADET_SPLIT_BUCKETS", "1") != "0":
            nblk = 1 + max(int(n[len(pf):].split(".")[0]) for n in conv_names[r[0]:r[1]])
            runs, hi = [], nblk - 1
            while hi >= 0:                         # runs of blocks [lo, hi], last block first, each >= min_bytes
                lo, size = hi, 0
                while True:
                    size += 4 * sum(e - b for _, b, e in flat.group_ranges([f"{pf}{lo}."]))
                    if size >= min_bytes or lo == 0:
                        break
                    lo -= 1
                runs.append((lo, hi))
                hi = lo - 1
            if len(runs) > 1 and 4 * sum(e - b for _, b, e in flat.group_ranges([f"{pf}{k}." for k in range(runs[-1][0], runs[-1][1] + 1)])) < min_bytes:
                runs[-2] = (runs[-1][0], runs[-2][1])      # a short remainder joins the run before it
                runs.pop()
            if len(runs) > 1:
                li = int(pf[len("backbone.layer"):-1]) - 1
                for lo, hi in runs:
                    names = [f"{pf}{k}." for k in range(lo, hi + 1)]
                    cr = [rng(n) for n in names]
                    buckets.append(bucket(f"{pf}{lo}-{hi}." if lo != hi else names[0], names, (min(c[0] for c in cr), max(c[1] for c in cr)),
                                          blocks=(li, lo, hi)))
                continue
        buckets.append(bucket(pf, ["backbone.conv1", "backbone.bn1"] if pf == "backbone.conv1" else [pf], r))
    return buckets


class _StreamWork:
    """What GradReducer needs of a c10d Work, for a collective issued synchronously on a stream of ours: an event behind it"""

    def __init__(self, timed, t0):
        self.t0 = t0
        self.done = torch.cuda.Event(enable_timing=timed)
        self.done.record()

    def wait(self):
        torch.cuda.current_stream().wait_event(self.done)

    def _get_duration(self):                                   # ms, valid once `done` has completed
        return self.t0.elapsed_time(self.done)


def sync_collectives_run_on_current_stream():
    """c10d issues `async_op=False` NCCL / RCCL collectives on the caller's current stream since torch 2.8 (before that
    every collective ran on the process group's internal stream and the caller's stream waited for it).  The gradient exchange
    relies on it to stay inside the four-stream budget (engine.py): on an older build the collectives would silently move
    to RCCL's own stream -- a fifth stream, measured +35 % step time."""
    try:
        major, minor = (int(x) for x in torch.__version__.split("+")[0].split(".")[:2])
    except ValueError:
        return True
    return (major, minor) >= (2, 8)


class GradReducer:
    """Sum-all-reduce of gradient-arena buckets as they become ready (RCCL over xGMI on the GPU box,
    gloo in the CPU tests).  `bucket_ready` is called on the stream that produced the bucket (the engine's side
    stream, right after the bucket's slab reduction); the collective runs behind an event of that stream on the
    communication stream (`comm_stream`: see __init__; without one, asynchronously on the process group's internal
    stream), so neither the side stream nor the main stream waits for it, and `finish()` makes the current stream wait
    for all of them.  The mean (1/world) is applied later, inside the fused clip+AdamW kernel."""

    def __init__(self, grads, device, bf16=False, comm_stream=None):
        """bf16=True: each bucket is exchanged as bf16 (half the bytes on xGMI; the reference's fp16 training also
        all-reduces half-precision gradients): fp32 -> bf16 staging buffer -> all-reduce -> back into the fp32 arena.
        comm_stream: a HIP stream of the caller on which the collectives run, as SYNCHRONOUS `all_reduce` calls (which
        c10d issues on the current stream) behind an event of the producing stream -- instead of asynchronous ones on the
        process group's internal stream.  The detector passes its tower-chain stream, idle from the end of the head's
        backward pass to the next step: a process may not USE more than four HIP streams on this device ("stream budget",
        engine.py), and main + side + second weight-gradient stream + chain are four already (with RCCL's own stream as the
        fifth the step took 13.8-14.1 instead of 10.2-10.3 ms, tools/bench_dp1.py)."""
        self.grads, self.device, self.bf16 = grads, device, bf16
        self.comm_stream = comm_stream if grads.is_cuda else None
        self.works = []
        self.staging = torch.empty(grads.numel(), dtype=torch.bfloat16, device=device) if bf16 else None
        # diagnostics (enable_trace()): per bucket an event on the producing stream when the bucket is handed over, an event
        # on the main stream right behind finish()'s wait for it ("done by": an upper bound of the completion, exact for the
        # bucket the optimizer really waits for) and -- with TORCH_NCCL_ENABLE_TIMING=1 -- RCCL's own start-to-end time of
        # the collective.  (Taking the completion on a stream of its own that only waits for the work was tried: that
        # stream's barrier packets stalled the main stream's hardware queue, +45 % step time under tracing.)
        self.trace = None

    def enable_trace(self, on=True):
        self.trace = [] if on else None

    def _convert(self, src, dst):
        if src.is_cuda:
            K.convert_rows(src.view(1, -1), dst.view(1, -1))
        else:
            dst.copy_(src)

    def bucket_ready(self, bucket):
        b, e = bucket["arena"]
        ready = None
        if self.trace is not None and self.grads.is_cuda:
            ready = torch.cuda.Event(enable_timing=True)
            ready.record()
        if self.comm_stream is not None:
            cs = self.comm_stream
            handed = torch.cuda.Event()
            handed.record()                                    # on the stream that produced the bucket
            timed = self.trace is not None
            with torch.cuda.stream(cs):
                cs.wait_event(handed)
                t0 = torch.cuda.Event(enable_timing=True) if timed else None
                if timed:
                    t0.record()
                if self.bf16:
                    st = self.staging[b:e]
                    self._convert(self.grads[b:e], st)
                    dist.all_reduce(st, op=dist.ReduceOp.SUM, async_op=False)
                else:
                    dist.all_reduce(self.grads[b:e], op=dist.ReduceOp.SUM, async_op=False)
                w = _StreamWork(timed, t0)
        elif self.bf16:
            st = self.staging[b:e]
            self._convert(self.grads[b:e], st)
            w = dist.all_reduce(st, op=dist.ReduceOp.SUM, async_op=True)
        else:
            w = dist.all_reduce(self.grads[b:e], op=dist.ReduceOp.SUM, async_op=True)
        self.works.append((w, b, e))
        if ready is not None:
            self.trace.append(dict(prefix=bucket["prefix"], bytes=(e - b) * (2 if self.bf16 else 4), ready=ready, done=None, work=w))

    def finish(self):
        for w, b, e in self.works:
            w.wait()
            if self.trace is not None and self.grads.is_cuda:
                for t in self.trace[-len(self.works):]:
                    if t["work"] is w and t["done"] is None:
                        t["done"] = torch.cuda.Event(enable_timing=True)
                        t["done"].record()
            if self.bf16:
                self._convert(self.staging[b:e], self.grads[b:e])
        self.works = []


class DetectorRuntime:
    def __init__(self, det, depth, num_classes, frozen_stages, strides, stacked_convs=4, math=None):
        dev = next(det.parameters()).device
        if dev.type != "cuda":
            raise _lib.RadetHipError(
                "radet_amd runs on MI355X only: move the detector to a HIP device (model.cuda()) first; "
                "there is no CPU / PyTorch fallback path.")
        _lib.load()
        self.dev = dev
        self.flat = FlatParams(det, dev)
        self.engine = Engine(self.flat.p, self.flat.g, depth=depth, num_classes=num_classes,
                             frozen_stages=frozen_stages, strides=strides, stacked_convs=stacked_convs, math=math,
                             watch=self.flat.watch)
        self.num_classes, self.strides = num_classes, tuple(strides)
        self.opt_state = None
        self.step_count = 0
        self.comm_stream = None
        self.buckets = compute_buckets(self.flat, [c.name for c in self.engine.convs],
                                       [c.trainable for c in self.engine.convs])
        self.reducer = None
        self._replicas_synced = False
        # (no collective here: a runtime may be built by a subset of the ranks -- rank-0 evaluation, checkpoint
        # conversion.  Data-parallel replicas are equalised by init_optimizer(), the entry point of every training run.)

    def caller_stream(self):
        """a HIP stream for the caller's own asynchronous work (next-batch upload) that stays off the main stream's pipe of the
        command processor (Engine.caller_stream; INTEGRATION.md, streams)"""
        return self.engine.caller_stream()

    def sync_replicas(self, src=0):
        """Broadcast the parameter arenas (trainable + frozen / BN statistics) and, once it exists, the optimizer
        state from rank `src`: what wrapping the model in MMDistributedDataParallel does at construction in the
        reference (radet/apis/train.py:73-81) and what a checkpoint loaded on one rank needs.  Collective: every rank
        of the default process group must call it.  No-op without an initialised group / with a single rank."""
        self._replicas_synced = True
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return False
        arenas = [self.flat.params, self.flat.frozen]
        if self.opt_state is not None:
            arenas += [self.opt_state["m"], self.opt_state["v"]]
        for t in arenas:
            dist.broadcast(t, src)
        step = torch.tensor([self.step_count], dtype=torch.int64, device=self.dev)
        dist.broadcast(step, src)
        self.step_count = int(step.item())
        return True

    # ------------------------------------------------------------------ inputs
    def pack_targets(self, gt_bboxes, gt_labels, points_to_gt_index, points_weight):
        """list-per-image tensors (the reference's DataContainer layout) -> flat device tensors."""
        dev = self.dev
        B = len(gt_bboxes)
        counts = [int(b.shape[0]) for b in gt_bboxes]
        off = np.zeros(B + 1, np.int32)
        off[1:] = np.cumsum(counts)
        boxes = torch.cat([b.reshape(-1, 4).float() for b in gt_bboxes]) if sum(counts) else torch.zeros(0, 4)
        labels = torch.cat([l.reshape(-1).long() for l in gt_labels]) if sum(counts) else torch.zeros(0, dtype=torch.long)
        boxes = boxes.to(dev).contiguous()
        labels = labels.to(dev).contiguous()
        if boxes.shape[0] == 0:   # keep valid device pointers
            boxes = torch.zeros(1, 4, device=dev)
            labels = torch.zeros(1, dtype=torch.long, device=dev)
        p2g = torch.stack([t.reshape(-1).long() for t in points_to_gt_index]).to(dev).contiguous()
        pw = torch.stack([t.reshape(-1).float() for t in points_weight]).to(dev).contiguous()
        return dict(boxes=boxes, labels=labels, off=torch.from_numpy(off).to(dev), p2g=p2g, pw=pw, B=B)

    # ------------------------------------------------------------------ forward / backward
    def forward(self, img, fold=True):
        """img: NCHW fp32 device tensor. Runs fold -> backbone -> neck -> head. Returns engine buffers."""
        assert img.is_cuda and img.dtype == torch.float32 and img.dim() == 4 and img.shape[1] == 3
        img = img.contiguous()
        e = self.engine
        e.prepare(img.shape[0], img.shape[2], img.shape[3])
        if fold:
            e.fold()
        feats = e.backbone_forward(img)
        P = e.neck_forward(feats)
        return e.head_forward(P)

    loss_hparams = None
    loss_weights = None      # device f32[3] = (loss_cls.loss_weight, 1, loss_iou.loss_weight) when either differs from 1

    def set_loss_from_head(self, head):
        """Take focal alpha / gamma and the three loss weights from the head's loss modules (config values)."""
        self.loss_hparams = dict(alpha=float(head.loss_cls.alpha), gamma=float(head.loss_cls.gamma),
                                 lbw=float(head.loss_bbox.loss_weight))
        wc, wi = float(head.loss_cls.loss_weight), float(head.loss_iou.loss_weight)
        self.loss_weights = None if (wc == 1.0 and wi == 1.0) else torch.tensor([wc, 1.0, wi], device=self.dev)
        return self

    def loss(self, tg, grad_scale=None, labels_out=None, tgt_out=None):
        hp = self.loss_hparams or dict(alpha=0.25, gamma=2.0, lbw=2.0)
        return self.engine.loss(tg["boxes"], tg["labels"], tg["off"], tg["p2g"], tg["pw"], grad_scale=grad_scale,
                                labels_out=labels_out, tgt_out=tgt_out, **hp)

    def backward(self, bucket_hook=None, next_img=None):
        """Reverse program. After each parameter group's wgrads are done its slabs are reduced /
        un-folded into the gradient arena and `bucket_hook(bucket)` may start its all-reduce.  next_img: the NEXT step's
        batch, if the caller already has it -- its frozen prefix (stem, frozen stages) then runs next to this backward pass
        (Engine.prefetch_prefix)."""
        e = self.engine
        if not self._replicas_synced:
            # the reference's DDP wrap broadcasts rank 0's parameters at construction (apis/train.py:73-81); here that is
            # sync_replicas(), called by init_optimizer() / load_checkpoint().  A module-API training loop with its own
            # torch optimizer never passes through those: say so once instead of training diverging replicas silently
            self._replicas_synced = True
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                import warnings
                warnings.warn("radet_amd: backward pass under a process group of "
                              f"{dist.get_world_size()} ranks before DetectorRuntime.sync_replicas() / init_optimizer() was "
                              "called: the replicas still hold their own initial parameters (call "
                              "det.runtime().sync_replicas() on every rank once, see INTEGRATION.md)")

        def unfold(bucket):
            self._unfold_bucket(bucket, bucket_hook)

        bk = {b["prefix"]: b for b in self.buckets}
        per_block = {(b["blocks"][0], b["blocks"][1]): b for b in self.buckets if "blocks" in b}   # keyed by the run's FIRST block
        dP = e.head_backward()
        if next_img is not None and bucket_hook is None:
            e.prefetch_prefix(next_img)         # on the tower-chain stream, idle from here on (single-GPU runs)
        unfold(bk["bbox_head."])
        d_feats = e.neck_backward(dP)
        unfold(bk["neck."])

        def after(li, bi=None):
            if bi is not None:                  # a block's weight gradients are all issued: its run's bucket, if it ends one
                if (li, bi) in per_block:
                    unfold(per_block[(li, bi)])
                return
            pf = "backbone.conv1" if li == "stem" else f"backbone.layer{li + 1}."
            if pf in bk:                        # (a stage that is cut into per-block buckets has none of its own)
                unfold(bk[pf])
        e.backbone_backward(d_feats, after_stage=after, block_ends=set(per_block))
        e.join_side()                           # gradients complete on the current stream from here on
        if next_img is not None and bucket_hook is not None:
            # data-parallel runs: the tower-chain stream carries the gradient exchange during the backward pass, so the prefix
            # follows the last bucket there (next to the exposed end of the exchange, the clip and AdamW)
            e.prefetch_prefix(next_img)

    def _unfold_bucket(self, bucket, bucket_hook=None):
        """Reduce / un-fold the weight-gradient slabs of one bucket's convs into the gradient arena."""
        import ctypes as C
        e = self.engine
        desc_bytes = C.sizeof(_lib.RadetConvDesc)
        a, b = bucket["convs"]
        if e.use_streams:
            # the slab reduction follows the bucket's weight-gradient GEMMs on the side stream, off the
            # critical path of the dgrad chain; the all-reduce hook keys off the side stream too
            side = e._side()
            e.side_collect()
            e._fork(side)
            with torch.cuda.stream(side):
                K.unfold_grads(e.table[a * desc_bytes:], b - a, e.max_cout)
                if bucket_hook is not None:
                    if _lib.TAPE is not None:       # a replayed step hands the bucket over at the same point, on this stream
                        _lib.TAPE.cut(lambda: bucket_hook(bucket), side)
                    bucket_hook(bucket)
        else:
            K.unfold_grads(e.table[a * desc_bytes:], b - a, e.max_cout)
            if bucket_hook is not None:
                if _lib.TAPE is not None:
                    _lib.TAPE.cut(lambda: bucket_hook(bucket), None)
                bucket_hook(bucket)

    # ------------------------------------------------------------------ optimiser
    def init_optimizer(self, lr=4e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05, max_norm=35.0, sync=True):
        """AdamW state for the trainable arena.  COLLECTIVE when a process group is initialised (sync=True): every rank
        must call it; the replicas then start from rank 0's parameters -- the broadcast MMDistributedDataParallel does at
        construction in the reference (radet/apis/train.py:73-81)."""
        n = self.flat.n_train
        self.opt_state = dict(m=torch.zeros(n, device=self.dev), v=torch.zeros(n, device=self.dev), lr=lr, betas=betas,
                              eps=eps, wd=weight_decay, max_norm=max_norm,
                              partials=torch.zeros(1024, device=self.dev), grad_norm=torch.zeros(1, device=self.dev))
        self.step_count = 0
        if sync:
            self.sync_replicas()

    def optimizer_step(self, lr=None, grad_div=1.0):
        st = self.opt_state
        self.step_count += 1
        K.sqnorm_partials(self.flat.grads, self.flat.n_train, st["partials"])
        K.adamw_step(self.flat.params, self.flat.grads, st["m"], st["v"], self.flat.n_train, st["lr"] if lr is None else lr,
                     st["betas"], st["eps"], st["wd"], self.step_count, st["max_norm"], grad_div, st["partials"],
                     st["grad_norm"])
        if _lib.TAPE is not None:                 # the two arguments of radet_adamw_step that move from step to step
            _lib.TAPE.mark("lr", 5)
            _lib.TAPE.mark("step", 10)
        self.engine.params_changed()              # the kernel wrote the arena behind torch's back: fold again next step

    def bf16_buckets(self):
        """Gradient buckets travel as bf16 (fp32 arena -> bf16 staging -> all-reduce -> back) by default in the bf16-STORAGE
        mode only: its step is half as long as the fp32 one while the fp32 exchange (127.7 MB per step, ~1.5 ms of per-link
        ring time on xGMI) stays the same, so the exchange is twice as exposed there; the mode's activation
        gradients are bf16 already, and the reference's fp16 training (apis/train.py:113-117) all-reduces half-precision
        gradients as well.  fp32 and the bf16-math mode (fp32 tensors) keep fp32 buckets.  RADET_BF16_BUCKETS=0 / 1 overrides."""
        env = os.environ.get("RADET_BF16_BUCKETS")
        if env is not None:
            return env == "1" and self.engine.math_name != "fp32"
        return self.engine.math_name == "bf16-storage"

    def comm_report(self):
        """Diagnostics of the traced steps since reducer.enable_trace() (averaged), times in ms after the start of the
        backward pass: per bucket `ready` (handed to RCCL: its slab reduction finished on the side stream), `allreduce_ms`
        (start-to-end time of the collective: HIP events around it on the communication stream, or RCCL's own timing with
        TORCH_NCCL_ENABLE_TIMING=1 when the process group's internal stream is used) and `done_by` (the main stream saw it complete in
        finish(): an upper bound, exact for the bucket that is waited for); `exposed_comm_ms` = how long the main stream
        waits for the exchange after its last backward kernel -- what clip + AdamW pays.  Host-synchronising: call it after
        the steps, not between them."""
        tr = self.reducer.trace if self.reducer is not None else None
        marks = self.__dict__.get("comm_marks")
        if not tr or not marks:
            return None
        torch.cuda.synchronize()
        n, lo = len(marks), 0
        acc = None
        fwd = bwd = exposed = 0.0
        for ev_b, ev0, ev1, hi in marks:
            step = tr[lo:hi]
            lo = hi
            def dur(w):                          # RCCL's own timing of the collective (TORCH_NCCL_ENABLE_TIMING=1), else -1
                try:
                    return float(w._get_duration())
                except Exception:
                    return -1.0
            rows = [(t["prefix"], t["bytes"], ev0.elapsed_time(t["ready"]), ev0.elapsed_time(t["done"]), dur(t["work"])) for t in step]
            b = ev0.elapsed_time(ev1)
            fwd += ev_b.elapsed_time(ev0); bwd += b
            exposed += max(0.0, max((r[3] for r in rows), default=0.0) - b)     # (a traced step may have handed over no bucket)
            acc = rows if acc is None else [(a[0], a[1], a[2] + r[2], a[3] + r[3], a[4] + r[4]) for a, r in zip(acc, rows)]
        rep = dict(steps=n, forward_loss_ms=round(fwd / n, 3), backward_ms=round(bwd / n, 3),
                   exposed_comm_ms=round(exposed / n, 3), bf16_buckets=bool(self.reducer.bf16),
                   buckets=[dict(bucket=a[0], mbytes=round(a[1] / 1e6, 2), ready_ms=round(a[2] / n, 3), done_by_ms=round(a[3] / n, 3),
                                 allreduce_ms=(round(a[4] / n, 3) if a[4] >= 0 else None)) for a in acc])
        self.reducer.trace.clear()
        self.comm_marks.clear()
        return rep

    # ------------------------------------------------------------------ data-parallel train step
    # ------------------------------------------------------------------ launch tape (radet_amd/tape.py)
    # RADET_TAPE=1 (default): once a train step has run `tape_after` times with the same geometry / mode, the next one is
    # recorded (every C-ABI call and cross-stream event operation it issues) and the steps after that are REPLAYED by one C
    # call per segment -- the same calls on the same streams, bit-identical results, ~5x less host time per step.
    # RADET_TAPE=0: always the eager step.  A step falls back to eager (and drops the tape) whenever something the tape baked
    # in has moved: another geometry plan, re-folded frozen weights, a changed optimizer hyper-parameter, tracing / per-kernel
    # timing switched on, a prefetch hand-over (next_img).
    tape_mode = os.environ.get("RADET_TAPE", "1")
    tape_after = int(os.environ.get("RADET_TAPE_AFTER", "2"))

    def _tape_key(self, img, tg, world, use_reducer):
        e, st = self.engine, self.opt_state
        return (e.geo_key, getattr(e, "plan_id", None), tuple(img.shape), img.dtype, tuple(tg["p2g"].shape), int(tg["off"].numel()),
                world, use_reducer, self.loss_weights is None or self.loss_weights.data_ptr(),
                st["betas"], st["eps"], st["wd"], st["max_norm"], st["m"].data_ptr(), e.wgrad_streams, e.use_streams,
                tuple(sorted(self.loss_hparams.items())) if self.loss_hparams else None)

    def drop_tape(self):
        self._tape = None

    _tape = None            # dict(key, count, tape, failed)

    def tape_stats(self):
        t = (self._tape or {}).get("tape")
        return None if t is None else dict(t.stats(), replays=t.replays)

    def train_step(self, img, tg, lr=None, next_img=None):
        """One optimisation step. With torch.distributed initialised (backend nccl = RCCL) gradients
        are summed across ranks per bucket on a side stream while the backward of earlier layers is
        still running; the mean (1/world) is folded into the fused clip+AdamW kernel.
        next_img: the batch of the NEXT call, if it is already on the device (a data loader one batch ahead): the frozen part
        of its forward pass (stem + frozen stages, `frozen_stages` of resnet.py:572-588) is then computed during this step's
        backward pass and picked up by the next call when it is given the same tensor; results are bit-identical to calls
        without it.
        Steady-state calls are replayed from a launch tape (see above); results are bit-identical to the eager step."""
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # a 1-rank process group still exercises the bucketed exchange when forced (single-GPU test of the RCCL path)
        use_reducer = world > 1 or (world == 1 and dist.is_available() and dist.is_initialized()
                                    and os.environ.get("RADET_FORCE_REDUCER") == "1")
        e = self.engine
        record = None
        tapeable = (self.tape_mode != "0" and next_img is None and K.EVENTS is None and e.tower_events is None
                    and not (self.reducer is not None and self.reducer.trace is not None) and e._pfx_ready is None
                    and img is not None and img.is_cuda and img.dtype == torch.float32 and img.is_contiguous()
                    and self.opt_state is not None and "partials" in self.opt_state
                    and all(tg[k].is_contiguous() for k in ("boxes", "labels", "off", "p2g", "pw")))
        if tapeable:
            key = self._tape_key(img, tg, world, use_reducer)
            st = self._tape
            if st is None or st["key"] != key:
                st = self._tape = dict(key=key, count=0, tape=None, failed=None)
            if st["tape"] is not None:
                if self._tape_still_valid():
                    return self._replay_step(st["tape"], img, tg, lr, world)
                st = self._tape = dict(key=key, count=0, tape=None, failed=None)      # something moved: start over
            st["count"] += 1
            if st["count"] > self.tape_after and st["failed"] is None and (not use_reducer or self.reducer is not None):
                from .tape import Tape
                record = Tape()
        if self.reducer is not None and self.reducer.trace is not None:
            self._ev_begin = torch.cuda.Event(enable_timing=True)
            self._ev_begin.record()
        if record is not None:
            torch.cuda.current_stream().synchronize()     # (recording compares allocator counters: nothing of ours in flight)
            record.begin(self.dev)
        try:
            self.forward(img)
            self.loss(tg, grad_scale=self.loss_weights)
            if use_reducer:
                if self.reducer is None:
                    on_chain = self.engine.use_streams and self.flat.grads.is_cuda
                    if on_chain and not sync_collectives_run_on_current_stream():
                        # the process group's internal stream will carry the exchange: keep the process at four streams by
                        # giving up the second weight-gradient stream (10.43 instead of 9.97 ms at one rank)
                        import warnings
                        warnings.warn(f"torch {torch.__version__} runs synchronous collectives on the process group's own stream: "
                                      "the gradient exchange uses that stream and the engine drops its second weight-gradient "
                                      "stream to stay inside the four-stream budget of this device")
                        on_chain = False
                        self.engine.wgrad_streams = 1
                    self.reducer = GradReducer(self.flat.grads, self.dev, bf16=self.bf16_buckets(),
                                               comm_stream=self.engine._chain_stream() if on_chain else None)
                tr = self.reducer.trace is not None
                if tr:
                    ev0 = torch.cuda.Event(enable_timing=True)
                    ev0.record()                     # (after forward + loss: the exchange can only overlap the backward pass)
                self.backward(self.reducer.bucket_ready, next_img=next_img)
                if tr:
                    ev1 = torch.cuda.Event(enable_timing=True)
                    ev1.record()                     # end of the backward pass on the main stream
                    self.__dict__.setdefault("comm_marks", []).append((self._ev_begin, ev0, ev1, len(self.reducer.trace)))
                if record is not None:
                    record.cut(self.reducer.finish, None)
                self.reducer.finish()
            else:
                self.backward(next_img=next_img)
            self.optimizer_step(lr=lr, grad_div=float(world))
        except BaseException:
            if record is not None:
                record.abort()
            raise
        if record is not None:
            record.end()
            st = self._tape
            if record.poisoned is None and (e._last_fold != (False, True) or e._pfx_ready is not None):
                record.poisoned = "the recorded step was not a steady-state step (frozen weights were folded in it)"
            if record.poisoned is None:
                record.bind("img", img)
                for k in ("boxes", "labels", "off", "p2g", "pw"):
                    record.bind(k, tg[k])
                st["tape"] = record
                st["nf"] = e._frozen_conv_count()
            else:
                st["failed"] = record.poisoned
                if os.environ.get("RADET_TAPE_STRICT") == "1":
                    raise _lib.RadetHipError(f"launch tape could not be recorded: {record.poisoned}")
        if self.loss_weights is not None:       # reported values carry the configured loss weights, like the reference's
            return self.engine.losses * self.loss_weights
        return self.engine.losses

    def _tape_still_valid(self):
        """what a recorded step baked in beyond its key: the frozen convs' folded weights (folded outside the tape whenever
        their sources move) and the arenas the module's parameters view"""
        e, st = self.engine, self._tape
        nf = st["nf"]
        if nf > 0 and e._part_version(0, nf) != e._folded[0]:
            return False
        return self.flat.still_bound() and _lib.TAPE is None

    def _replay_step(self, tape, img, tg, lr, world):
        e, st = self.engine, self.opt_state
        self.step_count += 1
        tape.replay(tensors=dict(img=img, boxes=tg["boxes"], labels=tg["labels"], off=tg["off"], p2g=tg["p2g"], pw=tg["pw"]),
                    values=dict(lr=st["lr"] if lr is None else lr, step=self.step_count))
        e.params_changed()
        if self.loss_weights is not None:
            return e.losses * self.loss_weights
        return e.losses


# ---------------------------------------------------------------------- autograd bridge (drop-in API)
class _DetectorLossFn(torch.autograd.Function):
    """losses = f(img, params): forward = engine forward + fused loss; backward = the engine's reverse
    program. Gradients come back as clones of the gradient-arena views (autograd accumulates them)."""

    @staticmethod
    def forward(ctx, rt, img, tg, weights, *params):
        rt.forward(img)
        losses = rt.loss(tg)
        ctx.rt, ctx.tg, ctx.weights = rt, tg, weights
        out = losses * weights
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g0, g1, g2):
        rt = ctx.rt
        gs = (torch.stack([g0.reshape(()), g1.reshape(()), g2.reshape(())]).float() * ctx.weights).contiguous()
        rt.loss(ctx.tg, grad_scale=gs)       # head-output gradients scaled by the upstream gradients
        rt.backward()
        grads = tuple(rt.flat.g[n].clone() for n in rt.flat.train_names)
        return (None, None, None, None) + grads


def _losses_autograd(self, img, gt_bboxes, gt_labels, points_to_gt_index, points_weight):
    det = self.owner()
    head = det.bbox_head
    tg = self.pack_targets(gt_bboxes, gt_labels, points_to_gt_index, points_weight)
    self.set_loss_from_head(head)
    weights = torch.tensor([head.loss_cls.loss_weight, 1.0, head.loss_iou.loss_weight], device=self.dev)
    named = dict(det.named_parameters())
    plist = [named[n] for n in self.flat.train_names]
    l0, l1, l2 = _DetectorLossFn.apply(self, img.to(self.dev), tg, weights, *plist)
    return dict(loss_cls=l0, loss_bbox=l1, loss_iou=l2)


DetectorRuntime.losses_autograd = _losses_autograd


def owner_runtime(module):
    ref = getattr(module, "_owner_ref", None)
    det = ref() if ref is not None else None
    if det is None:
        raise RuntimeError(f"{type(module).__name__} executes through its detector's MI355X runtime: build the "
                           "detector with build_detector(cfg).cuda() and call it (or detector.extract_feat)")
    return det.runtime()


def standalone_forward(module, part, x):
    rt = owner_runtime(module)
    if part == "backbone":
        return rt.backbone_api(x)
    return rt.neck_api(x)


def _neck_api(self, inputs):
    """inputs: NCHW C2..C5 (as returned by the backbone module). Returns NCHW P3..P7."""
    with torch.no_grad():
        e = self.engine
        B = inputs[0].shape[0]
        feats = [None]
        for i, x in enumerate(inputs[1:]):
            blk = e.stages[i + 1][-1]
            rows = e.buf[f"l{i + 2}.{len(e.stages[i + 1]) - 1}.out"]
            K.nchw_to_nhwc(x.to(self.dev).contiguous(), rows, B, x.shape[1], x.shape[2], x.shape[3])
            feats.append(rows)
            del blk
        P = e.neck_forward(feats)
        return tuple(_rows_to_nchw(self, P, e.plv, e.feat))


DetectorRuntime.neck_api = _neck_api


# ---------------------------------------------------------------------- inference + module-level API
def _detect(self, img, img_metas, test_cfg, rescale=False):
    """simple_test: forward -> per-level threshold/top-k/decode -> NMS, all on the GPU, batched over
    images; returns [(dets f32[K,5], labels i64[K])] per image on the device."""
    with torch.no_grad():
        self.forward(img.to(self.dev))
        hw, sf = _meta_tensors(self, img_metas, rescale)
        return _post_collect(self, *_post_launch(self, hw, sf, test_cfg))


def _detect_stream(self, batches, test_cfg, rescale=False):
    """`detect` over an iterable of (img, img_metas) batches with the host one batch behind the device: batch k + 1's
    device program (forward, decode, NMS) is enqueued BEFORE the host waits for batch k's detection counts, so the GPU
    never idles on the per-batch hand-over (what the reference's `single_gpu_test` loop -- apis/test.py -- pays per batch
    as a device synchronisation).  Yields exactly what `detect` returns, in order.  The small per-batch uploads (image
    sizes, scale factors) go through pinned memory and the counts come back behind an event on the stream that produced
    them: nothing in the loop synchronises the main stream.  Decode + NMS of batch k (a dozen launches of a few workgroups each, ~0.3 ms at
    batch 8) run on a stream of their own NEXT TO the forward pass of batch k + 1: the head outputs alternate between two
    sets of buffers, and the forward pass that reuses a set first waits for the post-processing that read it."""
    # (no stream of its own: a process should not create more than four HIP streams on this device -- engine.py, "stream
    # budget" -- and the training step's tower-chain stream is idle here: the inference pass runs its reg chain on `side`)
    post = self.engine._chain_stream()
    overlap = os.environ.get("RADET_POST_OVERLAP", "1") != "0"
    HEAD_OUT = ("cls", "reg_u", "iou")
    alts = {}                    # id(plan) -> (plan, the other set of head-output buffers)
    post_ev = [None, None]       # events "decode / NMS done" of the batch before last and of the last batch

    def fetch_counts(outs, pinned):     # on the current stream: the detection counts into pinned memory + "arrived" event
        pinned.copy_(outs[3], non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        return done

    def collect(rec):
        outs, done, pinned = rec[:3]    # (rec[3] keeps the batch's size / scale tensors alive until its decode has run)
        done.synchronize()                                      # the host waits for THIS batch only
        counts = pinned.numpy()
        ob, osc, ol, _ = outs
        return [(torch.cat([ob[i, :int(counts[i])], osc[i, :int(counts[i]), None]], -1), ol[i, :int(counts[i])])
                for i in range(ob.shape[0])]

    def upload(a):
        return torch.from_numpy(a).pin_memory().to(self.dev, non_blocking=True)

    swapped = {}                 # id(plan) -> the plan currently holds its second set

    def run():
        prev = None
        with torch.no_grad():
            for img, img_metas in batches:
                hw = upload(np.asarray([[float(m["img_shape"][0]), float(m["img_shape"][1])] for m in img_metas], np.float32))
                sf = upload(np.stack([np.asarray(m["scale_factor"], np.float32).reshape(4) for m in img_metas])) if rescale else None
                img = img.to(self.dev, non_blocking=True)
                if overlap:
                    e = self.engine
                    e.prepare(img.shape[0], img.shape[2], img.shape[3])
                    plan = e.buf
                    if id(plan) not in alts or alts[id(plan)][0] is not plan:
                        alts[id(plan)] = (plan, {k: torch.empty_like(plan[k]) for k in HEAD_OUT})
                    other = alts[id(plan)][1]
                    for k in HEAD_OUT:                               # this batch writes the set the batch before last wrote
                        plan[k], other[k] = other[k], plan[k]
                    swapped[id(plan)] = not swapped.get(id(plan), False)
                    if post_ev[0] is not None:
                        torch.cuda.current_stream().wait_event(post_ev[0])   # ... once that batch's decode / NMS have read it
                    self.forward(img)
                    fwd = torch.cuda.Event()
                    fwd.record()
                    with torch.cuda.stream(post):
                        post.wait_event(fwd)
                        outs = _post_launch(self, hw, sf, test_cfg)
                        ev = torch.cuda.Event()
                        ev.record()
                        pinned = torch.empty(outs[3].shape, dtype=outs[3].dtype).pin_memory()
                        done = fetch_counts(outs, pinned)
                    post_ev[0], post_ev[1] = post_ev[1], ev
                else:
                    self.forward(img)
                    outs = _post_launch(self, hw, sf, test_cfg)
                    ev = torch.cuda.Event()
                    ev.record()
                    pinned = torch.empty(outs[3].shape, dtype=outs[3].dtype).pin_memory()
                    with torch.cuda.stream(post):                   # (the copy must not queue behind the next batch's forward pass)
                        post.wait_event(ev)
                        done = fetch_counts(outs, pinned)
                rec = (outs, done, pinned, (hw, sf))
                if prev is not None:
                    yield collect(prev)
                prev = rec
            if prev is not None:
                yield collect(prev)

    try:
        yield from run()
    finally:
        # leave every plan with its own set (captured graphs and the training step hold pointers to it).  After a complete
        # run collect() has host-synchronised the last batch; an abandoned generator (break / exception mid-iteration) can
        # still have a batch's decode / NMS reading the head outputs and the shared post-processing workspaces on the chain
        # stream: whatever the caller enqueues next on this stream waits for it
        for ev in post_ev:
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
        for pid, odd in swapped.items():
            if odd and pid in alts:
                plan, other = alts[pid]
                for k in HEAD_OUT:
                    plan[k], other[k] = other[k], plan[k]


def _meta_tensors(self, img_metas, rescale):
    hw = torch.tensor([[float(m["img_shape"][0]), float(m["img_shape"][1])] for m in img_metas], device=self.dev)
    sf = None
    if rescale:
        sf = torch.tensor(np.stack([np.asarray(m["scale_factor"], np.float32).reshape(4) for m in img_metas]), device=self.dev)
    return hw, sf


def _post_launch(self, hw, sf, test_cfg, head_out=None):
    """Device part of the post-processing (no host synchronisation, no host->device copies: capturable in a hipGraph):
    decode + NMS launches on the engine's head outputs (or on `head_out` = dict(cls, reg, iou, scales, ldesc, nlvl, B,
    level_hw) row buffers handed in through the module API).  Returns the output tensors (boxes, scores, labels, counts)."""
    e = self.engine
    if head_out is None:
        head_out = dict(cls=e.buf["cls"], reg=e.buf["reg_u"], iou=e.buf["iou"], scales=e.scales_tensor(), ldesc=e.ldesc,
                        nlvl=e.nlvl, B=e.B, level_hw=e.plv.hw)
    ho = head_out
    B, nlvl = ho["B"], ho["nlvl"]
    nms_pre = int(test_cfg.get("nms_pre", -1))
    if nms_pre <= 0:
        nms_pre = max(h * w for h, w in ho["level_hw"]) * self.num_classes
    cap = nlvl * nms_pre
    if cap > 65536:
        raise NotImplementedError(f"nms_pre={nms_pre}: more than 65536 candidates per image exceed the NMS kernel's 16-bit positions")
    key = ("post", B, nlvl, nms_pre)
    posts = self.__dict__.setdefault("_posts", {})
    if key not in posts:
        # the NMS workspace holds a dense cap x cap / 64 x 8 B suppression mask per image (+ sort keys): 7 MiB per image at
        # the BOP configs' nms_pre = 1000, ~310 MB at nms_pre = 10000, 512 MiB at the 65536-candidate limit
        need = K.nms_ws_bytes(B, cap) + K.decode_ws_bytes(B, nlvl, nms_pre)
        limit = int(float(os.environ.get("RADET_POST_WS_LIMIT_GB", "16")) * 2 ** 30)
        if need > limit:
            raise MemoryError(f"decode + NMS workspaces for batch {B} with nms_pre={nms_pre} ({cap} candidates per image) need "
                              f"{need / 2 ** 30:.1f} GiB (dense suppression mask: cap^2 / 8 bytes per image); lower nms_pre or the "
                              f"batch, or raise RADET_POST_WS_LIMIT_GB (now {limit / 2 ** 30:.0f})")
        while posts and (len(posts) >= 8 or sum(v["nws"].numel() for v in posts.values()) + need > limit):
            posts.pop(next(iter(posts)))
        dev = self.dev
        posts[key] = dict(
            boxes=torch.empty(B, cap, 4, device=dev), scores=torch.empty(B, cap, device=dev),
            ctr=torch.empty(B, cap, device=dev), labels=torch.empty(B, cap, dtype=torch.long, device=dev),
            count=torch.zeros(B, dtype=torch.int32, device=dev), cscore=torch.empty(B, cap, device=dev),
            dws=torch.empty(K.decode_ws_bytes(B, nlvl, nms_pre), dtype=torch.uint8, device=dev),
            nws=torch.empty(K.nms_ws_bytes(B, cap), dtype=torch.uint8, device=dev),
            aux0=torch.zeros(B, cap, dtype=torch.long, device=dev), aux1=torch.zeros(B, cap, dtype=torch.long, device=dev))
    p = posts[key]
    K.decode_candidates(ho["cls"], ho["reg"], ho["iou"], ho["scales"], ho["ldesc"], nlvl, B, self.num_classes,
                        float(test_cfg["score_thr"]), nms_pre, hw, sf, p["boxes"], p["scores"], p["ctr"], p["labels"],
                        p["count"], p["dws"])
    ncfg = dict(test_cfg["nms"])
    typ = ncfg.get("type")
    max_per_img = int(test_cfg.get("max_per_img", 100))
    k = max_per_img if max_per_img > 0 else cap
    ob = torch.zeros(B, k, 4, device=self.dev)
    osc = torch.zeros(B, k, device=self.dev)
    ol = torch.zeros(B, k, dtype=torch.long, device=self.dev)
    oc = torch.zeros(B, dtype=torch.int32, device=self.dev)
    if typ in ("vote", "global_vote"):
        ctype, vtype = ncfg.get("cluster_score", "cls"), ncfg.get("vote_score", "iou")
        prod = None

        def pick(t):
            nonlocal prod
            if isinstance(t, (list, tuple)):
                if prod is None:
                    prod = p["cscore"]
                    torch.mul(p["scores"], p["ctr"], out=prod)   # elementwise product of two sigmoid outputs (plumbing)
                return prod
            return p["scores"] if t == "cls" else p["ctr"]
        K.nms(p["boxes"], pick(ctype), pick(vtype), p["labels"], p["count"], B, cap, K.NMS_MODES[typ],
              float(ncfg.get("iou_threshold", 0.6)), bool(ncfg.get("iou_enable", False)), float(ncfg.get("sigma", 0.025)),
              max_per_img, ob, osc, ol, oc, p["aux0"], p["aux1"], p["nws"])
    else:
        torch.mul(p["scores"], p["ctr"], out=p["cscore"])
        K.nms(p["boxes"], p["cscore"], p["cscore"], p["labels"], p["count"], B, cap, 3,
              float(ncfg.get("iou_threshold", 0.5)), False, 0.0, k, ob, osc, ol, oc, p["aux0"], p["aux1"], p["nws"])
    return ob, osc, ol, oc


def _post_collect(self, ob, osc, ol, oc):
    counts = oc.cpu().numpy()
    out = []
    for i in range(ob.shape[0]):
        kk = int(counts[i])
        out.append((torch.cat([ob[i, :kk], osc[i, :kk, None]], -1), ol[i, :kk]))
    return out


def _detect_graph(self, img, img_metas, test_cfg, rescale=False):
    """`detect` with the whole device program (weight folding, ~110 conv / GroupNorm launches on two streams, decode,
    the NMS pipeline) captured once per (batch, image size, test_cfg) in a hipGraph and replayed: single-image
    serving is otherwise bound by the host's launch rate (3.2 ms of Python + ctypes enqueue for ~1.7 ms of GPU work
    at batch 1).  Inputs are copied into the graph's static buffers; results are read back like `detect`."""
    with torch.no_grad():
        img = img.to(self.dev)
        key = (tuple(img.shape), bool(rescale), repr(sorted(dict(test_cfg).items(), key=str)))
        cache = self.__dict__.setdefault("_graphs", {})
        self.engine.prepare(img.shape[0], img.shape[2], img.shape[3])
        g = cache.get(key)
        if g is not None and g["plan"] is not self.engine.buf:    # the geometry plan was evicted and rebuilt: stale pointers
            g = None
        if g is None:
            hw, sf = _meta_tensors(self, img_metas, rescale)
            static_img = img.clone()
            for _ in range(2):                                  # warm-up: prepare(), autotune, lazy workspaces
                self.forward(static_img)
                _post_launch(self, hw, sf, test_cfg)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            cap_stream = self.engine._chain_stream()            # (an existing stream: see engine.py, "stream budget")
            K.splitk_ws_for(cap_stream)                         # allocate the stream's split-K workspace outside capture
            self.engine.invalidate_fold()                       # the weight fold is part of the graph: replays see
            with torch.cuda.graph(graph, stream=cap_stream):    # the parameters of the moment, like the eager path
                self.forward(static_img)
                outs = _post_launch(self, hw, sf, test_cfg)
            # the captured launches hold raw pointers into the decode / NMS workspaces: the record keeps them alive when
            # the `_posts` cache drops its reference
            # (... as do the head-output buffers: detect_stream alternates the plan's entries between two sets)
            g = cache[key] = dict(graph=graph, img=static_img, hw=hw, sf=sf, outs=outs, plan=self.engine.buf,
                                  keep=list(self.__dict__.get("_posts", {}).values())
                                  + [self.engine.buf[k] for k in ("cls", "reg_u", "iou")])
        else:
            hw, sf = _meta_tensors(self, img_metas, rescale)
            g["hw"].copy_(hw)
            if sf is not None:
                g["sf"].copy_(sf)
        g["img"].copy_(img)
        g["graph"].replay()
        return _post_collect(self, *g["outs"])


def _rows_to_nchw(self, rows, levels, ch):
    """Row-major NHWC level slices -> list of NCHW tensors (module-API outputs)."""
    outs = []
    for i, (h, w) in enumerate(levels.hw):
        r0, r1 = levels.level_rows(i)
        t = torch.empty(levels.B, ch, h, w, device=self.dev)
        K.nhwc_to_nchw(rows[r0:r1], t, levels.B, ch, h, w)
        outs.append(t)
    return outs


def _extract_feat_api(self, img):
    with torch.no_grad():
        img = img.to(self.dev).contiguous()
        e = self.engine
        e.prepare(img.shape[0], img.shape[2], img.shape[3])
        e.fold()
        P = e.neck_forward(e.backbone_forward(img))
        return tuple(_rows_to_nchw(self, P, e.plv, e.feat))


def _backbone_api(self, img):
    with torch.no_grad():
        img = img.to(self.dev).contiguous()
        e = self.engine
        e.prepare(img.shape[0], img.shape[2], img.shape[3])
        e.fold()
        feats = e.backbone_forward(img)
        outs = []
        for li, f in enumerate(feats):
            lv = e.stages[li][-1]["lout"]
            outs.append(_rows_to_nchw(self, f, lv, f.shape[1])[0])
        return tuple(outs)


def _head_forward_api(self, feats):
    """feats: tuple of NCHW P3..P7 (as produced by extract_feat). Returns (cls_scores, bbox_preds, iou_preds)
    lists of NCHW tensors, bbox_preds already Scale'd + ReLU'd like RADetHead.forward_single."""
    with torch.no_grad():
        e = self.engine
        B = feats[0].shape[0]
        P = e.buf["P"]
        for i, f in enumerate(feats):
            r0, r1 = e.plv.level_rows(i)
            K.nchw_to_nhwc(f.to(self.dev).contiguous(), P[r0:r1], B, e.feat, f.shape[2], f.shape[3])
        cls, reg_u, iou = e.head_forward(P)
        reg = torch.empty_like(reg_u)
        K.scale_relu(reg_u, e.scales_tensor(), reg, e.ldesc, e.nlvl, B)
        return (_rows_to_nchw(self, cls, e.plv, self.num_classes), _rows_to_nchw(self, reg, e.plv, 4),
                _rows_to_nchw(self, iou, e.plv, 1))


DetectorRuntime.detect = _detect
DetectorRuntime.detect_stream = _detect_stream
DetectorRuntime.detect_graph = _detect_graph


def _postprocess(self, img_metas, test_cfg, rescale):
    hw, sf = _meta_tensors(self, img_metas, rescale)
    return _post_collect(self, *_post_launch(self, hw, sf, test_cfg))


DetectorRuntime._postprocess = _postprocess
DetectorRuntime.extract_feat_api = _extract_feat_api
DetectorRuntime.backbone_api = _backbone_api
DetectorRuntime.head_forward_api = _head_forward_api


# ---------------------------------------------------------------------- RADetHead module API (loss / get_bboxes / forward_train)
def _lists_to_rows(self, tensors, ch):
    """list of NCHW level tensors -> (Levels, [R, ch] fp32 row buffer in the head's level-major order)"""
    B = int(tensors[0].shape[0])
    lv = K.Levels([(int(t.shape[2]), int(t.shape[3])) for t in tensors], B)
    rows = torch.empty(lv.rows, ch, device=self.dev)
    for i, t in enumerate(tensors):
        assert t.shape[0] == B and t.shape[1] == ch, (tuple(t.shape), B, ch)
        r0, r1 = lv.level_rows(i)
        K.nchw_to_nhwc(t.detach().to(self.dev, torch.float32).contiguous(), rows[r0:r1], B, ch, t.shape[2], t.shape[3])
    return lv, rows


class _HeadLossFn(torch.autograd.Function):
    """RADetHead.loss on NCHW head outputs (bbox_preds AFTER Scale + ReLU, as the reference passes them):
    fused targets + focal + GIoU + IoU-BCE kernel; backward = the same kernel with the upstream gradients folded in,
    gradients returned in the inputs' NCHW layout."""

    @staticmethod
    def forward(ctx, rt, tg, weights, nlv, *tensors):
        cls, reg, iou = tensors[:nlv], tensors[nlv:2 * nlv], tensors[2 * nlv:]
        lv, rc = _lists_to_rows(rt, cls, rt.num_classes)
        _, rr = _lists_to_rows(rt, reg, 4)
        _, ri = _lists_to_rows(rt, iou, 1)
        ctx.rt, ctx.tg, ctx.weights, ctx.lv = rt, tg, weights, lv
        ctx.rows = (rc, rr, ri)
        ctx.srcs = [t.device for t in tensors]
        losses = _HeadLossFn.run(ctx, None)[0]
        out = losses * weights
        return out[0], out[1], out[2]

    @staticmethod
    def run(ctx, grad_scale):
        rt, tg, lv = ctx.rt, ctx.tg, ctx.lv
        rc, rr, ri = ctx.rows
        R, C = lv.rows, rt.num_classes
        ldesc, nlvl = K.level_desc(lv, rt.strides)
        dev = rt.dev
        hp = rt.loss_hparams or dict(alpha=0.25, gamma=2.0, lbw=2.0)
        losses = torch.zeros(3, device=dev)
        dcls, dreg, diou = torch.empty(R, C, device=dev), torch.empty(R, 4, device=dev), torch.empty(R, 1, device=dev)
        ws = torch.zeros(K.head_loss_ws_ints(R), dtype=torch.int32, device=dev)
        K.head_loss(rc, rr, ri, torch.ones(nlvl, device=dev), tg["boxes"], tg["labels"], tg["off"], tg["p2g"], tg["pw"], ldesc,
                    nlvl, lv.B, C, hp["alpha"], hp["gamma"], hp["lbw"], 1e-6, grad_scale, losses, dcls, C, dreg, 4, diou, 1,
                    torch.zeros(nlvl, device=dev), ws, flags=1)
        return losses, dcls, dreg, diou

    @staticmethod
    def backward(ctx, g0, g1, g2):
        rt, lv = ctx.rt, ctx.lv
        gs = (torch.stack([g0.reshape(()), g1.reshape(()), g2.reshape(())]).to(rt.dev, torch.float32) * ctx.weights).contiguous()
        _, dcls, dreg, diou = _HeadLossFn.run(ctx, gs)
        grads = []
        for rows, ch in ((dcls, rt.num_classes), (dreg, 4), (diou, 1)):
            grads += _rows_to_nchw(rt, rows, lv, ch)
        return (None, None, None, None) + tuple(g.to(d) for g, d in zip(grads, ctx.srcs))


def _head_loss_api(self, cls_scores, bbox_preds, iou_preds, gt_bboxes, gt_labels, points_to_gt_index, points_weight):
    """RADetHead.loss (radet/models/dense_heads/radet_head.py:173-288): lists of NCHW tensors per level in,
    dict(loss_cls, loss_bbox, loss_iou) out, differentiable w.r.t. the three input lists."""
    head = self.owner().bbox_head
    assert len(cls_scores) == len(bbox_preds) == len(iou_preds)
    self.set_loss_from_head(head)
    tg = self.pack_targets(gt_bboxes, gt_labels, points_to_gt_index, points_weight)
    weights = torch.tensor([head.loss_cls.loss_weight, 1.0, head.loss_iou.loss_weight], device=self.dev)
    l0, l1, l2 = _HeadLossFn.apply(self, tg, weights, len(cls_scores), *cls_scores, *bbox_preds, *iou_preds)
    return dict(loss_cls=l0, loss_bbox=l1, loss_iou=l2)


def _get_bboxes_api(self, cls_scores, bbox_preds, centernesses, img_metas, cfg, rescale=False, with_nms=True):
    """ATSSHead.get_bboxes / RADetHead._get_bboxes_single (atss_head.py:325-387, radet_head.py:55-170): NCHW lists
    -> [(det_bboxes [k, 5], det_labels [k])] per image (device tensors)."""
    if not with_nms:
        raise NotImplementedError("get_bboxes(with_nms=False) only feeds test-time augmentation, which is out of scope "
                                  "(flip=False in the BOP configs)")
    assert len(cls_scores) == len(bbox_preds) == len(centernesses)
    with torch.no_grad():
        lv, rc = _lists_to_rows(self, cls_scores, self.num_classes)
        _, rr = _lists_to_rows(self, bbox_preds, 4)
        _, ri = _lists_to_rows(self, centernesses, 1)
        ldesc, nlvl = K.level_desc(lv, self.strides)
        # bbox_preds already carry Scale + ReLU: decode with unit scales (relu is idempotent on them)
        ho = dict(cls=rc, reg=rr, iou=ri, scales=torch.ones(nlvl, device=self.dev), ldesc=ldesc, nlvl=nlvl, B=lv.B,
                  level_hw=lv.hw)
        hw, sf = _meta_tensors(self, img_metas, rescale)
        return _post_collect(self, *_post_launch(self, hw, sf, cfg, head_out=ho))


class _HeadTrainFn(torch.autograd.Function):
    """RADetHead.forward_train on NCHW pyramid features: head forward + fused loss through the engine; backward =
    the engine's head reverse program (gradients for the head parameters and for the input features)."""

    @staticmethod
    def forward(ctx, rt, tg, weights, nfeat, *args):
        feats = args[:nfeat]
        e = rt.engine
        B = int(feats[0].shape[0])
        for i, f in enumerate(feats):
            r0, r1 = e.plv.level_rows(i)
            K.nchw_to_nhwc(f.detach().to(rt.dev, torch.float32).contiguous(), e.buf["P"][r0:r1], B, e.feat, f.shape[2], f.shape[3])
        e.head_forward(e.buf["P"])
        losses = rt.loss(tg)
        ctx.rt, ctx.tg, ctx.weights, ctx.nfeat = rt, tg, weights, nfeat
        ctx.srcs = [f.device for f in feats]
        ctx.pnames = [n for n in rt.flat.train_names if n.startswith("bbox_head.")]
        out = losses * weights
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g0, g1, g2):
        rt = ctx.rt
        e = rt.engine
        gs = (torch.stack([g0.reshape(()), g1.reshape(()), g2.reshape(())]).to(rt.dev, torch.float32) * ctx.weights).contiguous()
        rt.loss(ctx.tg, grad_scale=gs)
        dP = e.head_backward()
        rt._unfold_bucket({b["prefix"]: b for b in rt.buckets}["bbox_head."])
        e.join_side()
        dfe = [g.to(d) for g, d in zip(_rows_to_nchw(rt, dP, e.plv, e.feat), ctx.srcs)]
        return (None, None, None, None) + tuple(dfe) + tuple(rt.flat.g[n].clone() for n in ctx.pnames)


def _head_forward_train_api(self, x, img_metas, gt_bboxes, gt_labels, points_to_gt_index, points_weight, proposal_cfg=None):
    """RADetHead.forward_train (radet_head.py:32-52): x = NCHW P3..P7 -> losses (and proposals with proposal_cfg)."""
    if gt_labels is None:
        raise NotImplementedError("RADetHead.forward_train without gt_labels (RPN use) is not on the RADet path")
    det = self.owner()
    head = det.bbox_head
    e = self.engine
    B = int(x[0].shape[0])
    hw = [(int(f.shape[2]), int(f.shape[3])) for f in x]
    shape = (img_metas[0].get("batch_input_shape") or img_metas[0].get("pad_shape")) if img_metas else None
    H, W = (int(shape[0]), int(shape[1])) if shape is not None else (hw[0][0] * self.strides[0], hw[0][1] * self.strides[0])
    e.prepare(B, H, W)
    if list(e.plv.hw) != hw:
        raise ValueError(f"feature sizes {hw} do not belong to a {H}x{W} input (expected {list(e.plv.hw)})")
    e.fold()
    e._await_fold()
    self.set_loss_from_head(head)
    tg = self.pack_targets(gt_bboxes, gt_labels, points_to_gt_index, points_weight)
    weights = torch.tensor([head.loss_cls.loss_weight, 1.0, head.loss_iou.loss_weight], device=self.dev)
    named = dict(det.named_parameters())
    plist = [named[n] for n in self.flat.train_names if n.startswith("bbox_head.")]
    l0, l1, l2 = _HeadTrainFn.apply(self, tg, weights, len(x), *x, *plist)
    losses = dict(loss_cls=l0, loss_bbox=l1, loss_iou=l2)
    if proposal_cfg is None:
        return losses
    with torch.no_grad():
        hwt, sf = _meta_tensors(self, img_metas, False)
        return losses, _post_collect(self, *_post_launch(self, hwt, sf, proposal_cfg))


DetectorRuntime.head_loss_api = _head_loss_api
DetectorRuntime.get_bboxes_api = _get_bboxes_api
DetectorRuntime.head_forward_train_api = _head_forward_train_api
