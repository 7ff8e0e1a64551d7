"""Iteration-based training harness around the native train step.

Reproduces what the reference delegates to mmcv (radet/apis/train.py:87-169, configs/base/default_runtime.py:1-26):
IterBasedRunner loop, AdamW + grad-clip (fused on the GPU), OneCycle learning rate with mmcv's defaults
(div_factor 25, final_div_factor 1e4, linear anneal, two phases), text log every `interval` iterations (one 3-float
device->host read instead of the reference's four blocking all-reduces per iteration) and checkpoints whose
`state_dict` uses the reference's parameter names."""
import time

import torch
import torch.distributed as dist


class OneCycleLR:
    """mmcv OneCycleLrUpdaterHook (policy='OneCycle', anneal_strategy='linear', three_phase=False)."""

    def __init__(self, max_lr, total_steps, pct_start=0.3, anneal_strategy="linear", div_factor=25.0,
                 final_div_factor=1e4, three_phase=False):
        if anneal_strategy not in ("linear", "cos"):
            raise ValueError("anneal_strategy must be 'cos' or 'linear'")
        if three_phase:
            raise NotImplementedError("three_phase=True is not used by the RADet configs")
        self.max_lr, self.total_steps, self.anneal = float(max_lr), int(total_steps), anneal_strategy
        self.initial_lr = self.max_lr / div_factor
        self.min_lr = self.initial_lr / final_div_factor
        self.up_end = float(pct_start * total_steps) - 1
        self.down_end = float(total_steps) - 1

    def _anneal(self, start, end, pct):
        if self.anneal == "linear":
            return (end - start) * pct + start
        import math
        return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1)

    def get_lr(self, step):
        """Learning rate used for iteration `step` (0-based), exactly torch.optim.lr_scheduler.OneCycleLR's value."""
        if step > self.down_end:
            raise ValueError(f"step {step} exceeds total_steps {self.total_steps}")
        if step <= self.up_end:
            return self._anneal(self.initial_lr, self.max_lr, step / self.up_end)
        return self._anneal(self.max_lr, self.min_lr, (step - self.up_end) / (self.down_end - self.up_end))


def optimizer_state_dict(model, runtime):
    """The fused AdamW's state in `torch.optim.AdamW.state_dict()` format ({state, param_groups}; parameter index =
    position in `model.parameters()`, what mmcv's `build_optimizer` hands to the optimizer), so checkpoints written
    here resume under the reference's runner and vice versa."""
    st = runtime.opt_state
    flat = runtime.flat
    names = [n for n, _ in model.named_parameters()]
    state = {}
    for i, n in enumerate(names):
        if n in flat.offsets:
            o, k = flat.offsets[n], flat.p[n].numel()
            state[i] = dict(step=torch.tensor(float(runtime.step_count)),
                            exp_avg=st["m"][o:o + k].view(flat.p[n].shape).cpu().clone(),
                            exp_avg_sq=st["v"][o:o + k].view(flat.p[n].shape).cpu().clone())
    group = dict(lr=st["lr"], betas=tuple(st["betas"]), eps=st["eps"], weight_decay=st["wd"], amsgrad=False,
                 params=list(range(len(names))))
    return dict(state=state, param_groups=[group])


def load_optimizer_state_dict(model, runtime, osd):
    """Accepts the torch / mmcv format above and the flat-arena format of earlier checkpoints of this package."""
    st = runtime.opt_state
    if "state" in osd and "param_groups" in osd:
        flat = runtime.flat
        names = [n for n, _ in model.named_parameters()]
        order = [i for g in osd["param_groups"] for i in g["params"]]
        if len(order) != len(names):
            raise ValueError(f"optimizer state has {len(order)} parameters, the model {len(names)}")
        step = 0
        for pos, key in enumerate(order):
            entry = osd["state"].get(key)
            n = names[pos]
            if entry is None or n not in flat.offsets:
                continue
            o, k = flat.offsets[n], flat.p[n].numel()
            st["m"][o:o + k].copy_(entry["exp_avg"].reshape(-1))
            st["v"][o:o + k].copy_(entry["exp_avg_sq"].reshape(-1))
            step = max(step, int(float(entry["step"])))
        runtime.step_count = step
    elif "step" in osd:                                   # flat arenas (round-1 checkpoints of this package)
        if osd["exp_avg"].numel() != st["m"].numel():
            raise ValueError("flat optimizer state does not match this model's parameter arena")
        runtime.step_count = int(osd["step"])
        st["m"].copy_(osd["exp_avg"])
        st["v"].copy_(osd["exp_avg_sq"])
    else:
        raise KeyError("unrecognised optimizer state: expected torch's {state, param_groups} or {step, exp_avg, exp_avg_sq}")


def save_checkpoint(model, path, meta=None, runtime=None):
    """mmcv-style checkpoint: dict(meta=..., state_dict=..., optimizer=...) with reference parameter names; the
    optimizer entry is `torch.optim.AdamW.state_dict()`-compatible."""
    ckpt = dict(meta=dict(meta or {}), state_dict={k: v.detach().cpu() for k, v in model.state_dict().items()})
    if runtime is not None and runtime.opt_state is not None:
        ckpt["optimizer"] = optimizer_state_dict(model, runtime)
    torch.save(ckpt, path)
    return path


def load_checkpoint(model, path, strict=False, runtime=None, sync=True):
    """Load a reference-format checkpoint.  With `runtime` and an initialised process group this is COLLECTIVE (sync=True):
    every rank must call it, and all continue from rank 0's parameters / optimizer state / step counter -- so a file that
    only rank 0 can read reaches every replica.  Pass sync=False for rank-local use (rank-0 evaluation, conversion)."""
    ckpt = torch.load(path, map_location="cpu")
    sd = ckpt.get("state_dict", ckpt)
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    missing = model.load_state_dict(sd, strict=strict)      # copies in place: parameters stay in the flat arena
    if runtime is not None and "optimizer" in ckpt and runtime.opt_state is not None:
        load_optimizer_state_dict(model, runtime, ckpt["optimizer"])
    if runtime is not None and sync:
        runtime.sync_replicas()                              # data-parallel: every rank continues from rank 0's state
    return ckpt.get("meta", {}), missing


def wrap_fp16_model(model, mode="bf16-storage"):
    """Mixed-precision switch with mmcv's name (the reference: `wrap_fp16_model` + `Fp16OptimizerHook`,
    apis/train.py:113-117, which keep half-precision activations and fp32 master weights).  Marks the detector so that
    its HIP runtime is built in `mode`:
      "bf16-storage" (default): bf16 activations / folded weights / activation gradients in HBM, bf16 matrix cores,
                                fp32 accumulate, GroupNorm statistics, loss, weight gradients, master weights, AdamW;
      "bf16":                   fp32 tensors, operands rounded to bf16 on their way into the matrix cores.
    No loss scaling is needed with bf16's fp32-sized exponent."""
    assert mode in ("bf16-storage", "bf16")
    model.fp16_enabled = True
    model.fp16_mode = mode
    return model


def train_detector(model, batches, cfg, max_iters=None, log=print, checkpoint_path=None):
    """`batches`: iterable of dict(img=f32[B,3,H,W], gt_bboxes, gt_labels, points_to_gt_index, points_weight)
    (device or host tensors; lists per image).  One process per GPU; gradients are averaged over the
    `torch.distributed` group if one is initialised.  Returns the list of logged loss triples."""
    if cfg.get("fp16", None) is not None:        # reference: Fp16OptimizerHook + wrap_fp16_model (apis/train.py:113-117)
        wrap_fp16_model(model)                   # here: bf16 operands, fp32 accumulate / master weights, no loss scaling
    rt = model.train().runtime()
    o = cfg.optimizer
    clip = cfg.get("optimizer_config", {}).get("grad_clip") or {}
    rt.init_optimizer(lr=o.lr, betas=tuple(o.betas), eps=o.eps, weight_decay=o.weight_decay,
                      max_norm=float(clip.get("max_norm", 0.0)))
    rt.set_loss_from_head(model.bbox_head)
    lc = cfg.lr_config
    sched = OneCycleLR(lc.max_lr, lc.total_steps, pct_start=lc.get("pct_start", 0.3),
                       anneal_strategy=lc.get("anneal_strategy", "cos"))
    max_iters = max_iters or cfg.runner.max_iters
    interval = cfg.get("log_config", {}).get("interval", 50)
    ckpt_every = cfg.get("checkpoint_config", {}).get("interval", 0)
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    history, t0 = [], time.time()
    for it, batch in enumerate(batches):
        if it >= max_iters:
            break
        lr = sched.get_lr(it)
        tg = rt.pack_targets(batch["gt_bboxes"], batch["gt_labels"], batch["points_to_gt_index"], batch["points_weight"])
        losses = rt.train_step(batch["img"].to(rt.dev), tg, lr=lr)
        if (it + 1) % interval == 0 or it + 1 == max_iters:
            vals = torch.cat([losses, rt.opt_state["grad_norm"]]).cpu().tolist()     # the only host sync
            history.append(vals[:3])
            if rank == 0:
                log(f"Iter [{it + 1}/{max_iters}] lr: {lr:.3e}, loss_cls: {vals[0]:.4f}, loss_bbox: {vals[1]:.4f}, "
                    f"loss_iou: {vals[2]:.4f}, loss: {sum(vals[:3]):.4f}, grad_norm: {vals[3]:.4f}, "
                    f"time: {(time.time() - t0) / (it + 1):.4f} s/iter")
        if checkpoint_path and ckpt_every and (it + 1) % ckpt_every == 0 and rank == 0:
            save_checkpoint(model, checkpoint_path.format(iter=it + 1), meta=dict(iter=it + 1), runtime=rt)
    return history
