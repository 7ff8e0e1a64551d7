"""Make the reference (/root/reference, mmdetection-2.8 fork) importable in THIS container.

Only used by tests/golden/gen_golden.py to produce the committed fixtures; it is
never imported by the product or by the test-suite itself (the reference does not
exist on the GPU box).  mmcv / cv2 / pycocotools / terminaltables are not installed,
so the handful of mmcv symbols the hot path executes are provided here as thin
wrappers over torch.nn (all arithmetic stays in torch or in the reference's own
code); everything else resolves to inert placeholder classes.
"""
import importlib.util
import logging
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"
REPO_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class _Placeholder:
    def __init__(self, *a, **k):
        pass

    def __init_subclass__(cls, **k):
        pass


class _Stub(types.ModuleType):
    """Module whose unknown attributes become placeholder classes."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        obj = type(name, (_Placeholder,), {})
        setattr(self, name, obj)
        return obj


def _mod(name, **attrs):
    m = _Stub(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent:
        setattr(sys.modules[parent], child, m)
    return m


# ---------------------------------------------------------------- mmcv.utils
class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            self._module_dict[name or cls.__name__] = cls
            return cls
        if module is not None:
            return _reg(module)
        return _reg


def build_from_cfg(cfg, registry, default_args=None):
    args = dict(cfg)
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop("type")
    cls = registry.get(obj_type) if isinstance(obj_type, str) else obj_type
    if cls is None:
        raise KeyError(f"{obj_type} is not in the {registry._name} registry")
    return cls(**args)


def print_log(msg, logger=None, level=logging.INFO):
    pass


def get_logger(name, log_file=None, log_level=logging.INFO):
    return logging.getLogger(name)


def is_tuple_of(seq, expected_type):
    return isinstance(seq, tuple) and all(isinstance(x, expected_type) for x in seq)


# ---------------------------------------------------------------- mmcv.cnn
def constant_init(module, val, bias=0):
    if hasattr(module, "weight") and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def kaiming_init(module, a=0, mode="fan_out", nonlinearity="relu", bias=0, distribution="normal"):
    if distribution == "uniform":
        nn.init.kaiming_uniform_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    else:
        nn.init.kaiming_normal_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def normal_init(module, mean=0, std=1, bias=0):
    nn.init.normal_(module.weight, mean, std)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def xavier_init(module, gain=1, bias=0, distribution="normal"):
    if distribution == "uniform":
        nn.init.xavier_uniform_(module.weight, gain=gain)
    else:
        nn.init.xavier_normal_(module.weight, gain=gain)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def uniform_init(module, a=0, b=1, bias=0):
    nn.init.uniform_(module.weight, a, b)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def bias_init_with_prob(prior_prob):
    return float(-np.log((1 - prior_prob) / prior_prob))


def build_conv_layer(cfg, *args, **kwargs):
    assert cfg is None or cfg.get("type", "Conv2d") in ("Conv", "Conv2d")
    return nn.Conv2d(*args, **kwargs)


def build_norm_layer(cfg, num_features, postfix=""):
    cfg = dict(cfg)
    t = cfg.pop("type")
    requires_grad = cfg.pop("requires_grad", True)
    cfg.setdefault("eps", 1e-5)
    if t == "BN":
        layer, abbr = nn.BatchNorm2d(num_features, **cfg), "bn"
    elif t == "GN":
        layer, abbr = nn.GroupNorm(num_channels=num_features, **cfg), "gn"
    else:
        raise KeyError(t)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return abbr + str(postfix), layer


class ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias="auto", conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"),
                 inplace=True, with_spectral_norm=False, padding_mode="zeros",
                 order=("conv", "norm", "act")):
        super().__init__()
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == "auto":
            bias = not self.with_norm
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride,
                              padding=padding, dilation=dilation, groups=groups, bias=bias)
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            assert act_cfg["type"] == "ReLU"
            self.activate = nn.ReLU(inplace=inplace)
        kaiming_init(self.conv, a=0, nonlinearity="relu")
        if self.with_norm:
            constant_init(getattr(self, self.norm_name), 1, bias=0)

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = getattr(self, self.norm_name)(x)
        if self.with_activation:
            x = self.activate(x)
        return x


class Scale(nn.Module):
    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        return x * self.scale


# ---------------------------------------------------------------- mmcv.runner
def _identity_decorator(*dargs, **dkwargs):
    def deco(fn):
        return fn
    return deco


def load_checkpoint(*a, **k):
    return {}


def get_dist_info():
    return 0, 1


def sigmoid_focal_loss_cpu(pred, target, gamma, alpha, weight, reduction):
    """mmcv.ops.sigmoid_focal_loss has no CPU kernel; elementwise formula of mmcv 1.3.18
    (`-t*a*(1-p)^g*log p - (1-t)(1-a)p^g log(1-p)`, background row = label C)."""
    assert weight is None and reduction == "none"
    n, c = pred.shape
    t = torch.zeros_like(pred)
    fg = (target >= 0) & (target < c)
    t[fg.nonzero(as_tuple=False).reshape(-1), target[fg]] = 1.0
    p = pred.sigmoid()
    flt_min = torch.finfo(torch.float32).tiny
    term_p = -alpha * (1 - p).pow(gamma) * torch.log(p.clamp(min=flt_min))
    term_n = -(1 - alpha) * p.pow(gamma) * torch.log((1 - p).clamp(min=flt_min))
    return t * term_p + (1 - t) * term_n


def install():
    """Install the fake third-party modules, then import the reference."""
    if "radet" in sys.modules:
        return sys.modules["radet"]
    _mod("mmcv", __version__="1.3.18")
    _mod("mmcv.utils", Registry=Registry, build_from_cfg=build_from_cfg, print_log=print_log,
         get_logger=get_logger, is_tuple_of=is_tuple_of)
    _mod("mmcv.cnn", ConvModule=ConvModule, Scale=Scale, build_conv_layer=build_conv_layer,
         build_norm_layer=build_norm_layer, constant_init=constant_init, kaiming_init=kaiming_init,
         normal_init=normal_init, xavier_init=xavier_init, uniform_init=uniform_init,
         bias_init_with_prob=bias_init_with_prob)
    _mod("mmcv.cnn.bricks")
    _mod("mmcv.cnn.bricks.transformer")
    _mod("mmcv.cnn.bricks.registry")
    _mod("mmcv.runner", force_fp32=_identity_decorator, auto_fp16=_identity_decorator,
         load_checkpoint=load_checkpoint, get_dist_info=get_dist_info)
    _mod("mmcv.runner.base_module")
    _mod("mmcv.ops")
    _mod("mmcv.ops.nms")
    _mod("mmcv.ops.roi_align")
    _mod("mmcv.parallel")
    _mod("mmcv.image")
    _mod("mmcv.onnx")
    _mod("mmcv.onnx.symbolic")
    _mod("mmcv.fileio")
    _mod("cv2")
    _mod("pycocotools", __version__="12.0.2")
    _mod("pycocotools.mask")
    _mod("pycocotools.coco")
    _mod("pycocotools.cocoeval")
    _mod("terminaltables")

    # the reference's own compiled ops (oracle/_ref), pre-seeded where the wrappers expect them
    sys.path.insert(0, REPO_ROOT)
    from oracle import build_ref
    build_ref.build()
    sys.modules["radet.ops.vote.vote_ext"] = build_ref.load("ref_vote_ext")
    sys.modules["radet.ops.cluster.cluster_ext"] = build_ref.load("ref_cluster_ext")
    _mod_b2d = _Stub("radet.ops.bbox2distance.bbox2distance_ext")
    sys.modules["radet.ops.bbox2distance.bbox2distance_ext"] = _mod_b2d

    sys.path.insert(0, REFERENCE_ROOT)
    import radet  # noqa: F401
    import radet.models  # noqa: F401
    import radet.core  # noqa: F401
    import radet.datasets.pipelines  # noqa: F401
    import radet.models.losses.focal_loss as fl
    fl._sigmoid_focal_loss = sigmoid_focal_loss_cpu
    return radet


class AttrDict(dict):
    """train_cfg / test_cfg container: attribute access, .copy() keeps the subclass."""
    __getattr__ = dict.get

    def copy(self):
        return AttrDict(self)


def attrify(d):
    if isinstance(d, dict):
        return AttrDict({k: attrify(v) for k, v in d.items()})
    if isinstance(d, (list, tuple)):
        return type(d)(attrify(v) for v in d)
    return d


def load_cfg(name="configs/bop/r50_ycbv_pbr.py"):
    ns = {}
    with open(os.path.join(REFERENCE_ROOT, name)) as f:
        exec(f.read(), ns)
    model = dict(ns["model"])
    model["pretrained"] = None
    return model, attrify(ns["train_cfg"]), attrify(ns["test_cfg"])
