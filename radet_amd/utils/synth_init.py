"""Seeded synthetic "trained-like" parameters for benchmarking (bench.py --weights synth).

The default initialisation (kaiming weights, BatchNorm / GroupNorm at identity, zero biases) on synthetic images gives
smooth, strongly correlated activations.  The conv GEMMs of the default fp32 arithmetic run against the chip's power
envelope, so their speed depends on how much the operand bits toggle (307 us on zero activations, 394 us on
dense Gaussian ones for the same launch).  This fill draws every parameter and running statistic from fixed, seeded
distributions of the spread trained detectors show, so that the activations are dense and decorrelated: the second value
in the bench line is the throughput on such data.  Product-side code: no oracle involved."""
import math

import torch


def synth_fill(module, seed=0):
    g = torch.Generator().manual_seed(int(seed))
    with torch.no_grad():
        for name, p in module.named_parameters():
            t = torch.empty(p.shape)
            if p.dim() == 4:                                     # conv weight: He-scaled normal, mild per-filter gain
                fan_in = p.shape[1] * p.shape[2] * p.shape[3]
                t.normal_(0.0, math.sqrt(2.0 / fan_in), generator=g)
                t *= (0.75 + 0.5 * torch.rand(p.shape[0], 1, 1, 1, generator=g))
            elif name.endswith(".scale"):                       # per-level Scale of the box branch
                t.fill_(1.0)
            elif name.endswith("weight"):                       # BN / GN gamma
                t.uniform_(0.5, 1.5, generator=g)
            else:                                               # biases, BN / GN beta
                t.normal_(0.0, 0.1, generator=g)
                if "atss_cls" in name:
                    t -= 4.6                                    # prior probability 0.01, as bias_init_with_prob
            p.copy_(t.to(p.device))
        for name, b in module.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(torch.empty(b.shape).normal_(0.0, 0.1, generator=g).to(b.device))
            elif name.endswith("running_var"):
                b.copy_(torch.empty(b.shape).uniform_(0.5, 1.5, generator=g).to(b.device))
    return module
