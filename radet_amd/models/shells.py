"""Parameter containers with the reference's sub-module / parameter names (state-dict compatible,
SURVEY.md §8b).  They hold torch Parameters and initialise them like the reference; they never
compute -- all arithmetic runs in the HIP engine (radet_amd/engine.py)."""
import math

import torch
from torch import nn


class ConvShell(nn.Module):
    """Holds `weight` [Cout,Cin,k,k] (+ `bias`) exactly like nn.Conv2d does in the reference."""

    def __init__(self, cin, cout, k, stride=1, padding=0, bias=False):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding = cin, cout, k, stride, padding
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        kaiming_normal_(self)

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}"


class BNShell(nn.Module):
    """BatchNorm2d parameters/buffers; always evaluated with running statistics (norm_eval=True)."""

    def __init__(self, c, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = c, eps
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class GNShell(nn.Module):
    def __init__(self, groups, c, eps=1e-5):
        super().__init__()
        self.num_groups, self.num_channels, self.eps = groups, c, eps
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class ConvModuleShell(nn.Module):
    """mmcv ConvModule naming: `.conv` and optionally `.gn`."""

    def __init__(self, cin, cout, k, stride=1, padding=0, gn_groups=None):
        super().__init__()
        self.conv = ConvShell(cin, cout, k, stride, padding, bias=gn_groups is None)
        if gn_groups is not None:
            self.gn = GNShell(gn_groups, cout)


class Scale(nn.Module):
    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))


def kaiming_normal_(m, mode="fan_out"):
    nn.init.kaiming_normal_(m.weight, a=0, mode=mode, nonlinearity="relu")
    if m.bias is not None:
        nn.init.constant_(m.bias, 0)


def xavier_uniform_(m):
    nn.init.xavier_uniform_(m.weight, gain=1)
    if m.bias is not None:
        nn.init.constant_(m.bias, 0)


def normal_(m, std=0.01, bias=0.0):
    nn.init.normal_(m.weight, 0, std)
    if m.bias is not None:
        nn.init.constant_(m.bias, bias)


def bias_init_with_prob(p):
    return float(-math.log((1 - p) / p))
