"""CPU tests of the host side: C ABI surface, registry / config drop-in, parameter inventory, geometry,
loud failure without a GPU.  No kernel is launched here."""
import ctypes
import math
import os
import re

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(REPO, "configs", "bop", "r50_ycbv_pbr.py")


def header_functions():
    src = open(os.path.join(REPO, "include", "radet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(radet_[a-z0-9_]+)\s*\(", src)))


def test_cabi_exports_every_declared_symbol():
    from radet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    names = header_functions()
    assert len(names) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/radet_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, set(_lib.SIGNATURES) ^ set(names)
    _lib.load()
    assert ctypes.sizeof(_lib.RadetConvDesc) == 200                 # 15 pointers + 9 32-bit fields (padded to 8) + 5 pointers
    assert ctypes.sizeof(_lib.RadetScales) == 96                    # 12 device pointers


def test_tape_thunks_are_current_and_replay_runs_host_calls():
    """The launch tape's generated thunks (radet_amd/csrc/tape_thunks.c) match _lib.SIGNATURES; radet_tape_replay unpacks
    the argument words of a recorded call correctly (checked on entry points that fail on bad arguments before they touch
    the device: the failing op's index and code come back) and the recorder produces the words ctypes would pass."""
    import subprocess
    import sys
    assert subprocess.run([sys.executable, os.path.join(REPO, "tools", "gen_tape_thunks.py"), "--check"]).returncode == 0
    from radet_amd import _lib
    from radet_amd.tape import Tape, _raw, _fbits
    lib = _lib.load()
    assert lib.radet_tape_fn_index(b"radet_adamw_step") >= 0 and lib.radet_tape_fn_index(b"radet_tape_replay") == -1
    assert ctypes.sizeof(_lib.RadetTapeOp) == 8 + 16 + 32 * 8
    arr = (ctypes.c_int * 3)(1, 2, 3)
    sc = _lib.RadetScales()
    assert _raw(None, ctypes.c_void_p)[0] == 0 and _raw(arr, ctypes.c_void_p)[0] == ctypes.addressof(arr)
    assert _raw(ctypes.byref(sc), ctypes.c_void_p)[0] == ctypes.addressof(sc)
    assert _raw(ctypes.c_size_t(7), ctypes.c_size_t)[0] == 7 and _raw(-3, ctypes.c_int)[0] == 2 ** 64 - 3
    assert _raw(0.25, ctypes.c_float)[0] == _fbits(0.25) == 0x3E800000
    ops = (_lib.RadetTapeOp * 3)()
    ops[0].kind, ops[0].fn = 0, lib.radet_tape_fn_index(b"radet_fill_zero")        # (NULL, 0 bytes): a no-op that succeeds
    ops[1].kind, ops[1].fn = 0, lib.radet_tape_fn_index(b"radet_copy_d2d")         # (NULL, NULL, 16 bytes): -1 before any HIP call
    ops[1].args[2] = 16
    ops[2].kind = 7                                                                 # unknown kind
    failed = ctypes.c_int(-1)
    assert lib.radet_tape_replay(ops, 0, 1, ctypes.byref(failed)) == 0 and failed.value == -1
    assert lib.radet_tape_replay(ops, 0, 3, ctypes.byref(failed)) == -1 and failed.value == 1
    assert lib.radet_tape_replay(ops, 2, 3, ctypes.byref(failed)) == -1 and failed.value == 2
    assert lib.radet_tape_replay(None, 0, 0, None) == -1
    t = Tape()
    t._ops.append((0, ops[0].fn, None, None, [0, 0, 0]))
    t._last_call = (0, _lib.SIGNATURES["radet_fill_zero"][1])
    t.mark("n", 1)
    t._alloc0, t.dev = 0, None
    t._alloc_count = lambda dev: 0
    _lib.TAPE = t
    t.end()
    assert _lib.TAPE is None and t.n == 1 and t.stats()["calls"] == 1
    t.replay(values=dict(n=0))
    assert t.replays == 1 and t.ops[0].args[1] == 0


def test_conv_geometry_refuses_tensors_beyond_the_kernels_address_range():
    """The tile loads address a tensor with 32-bit byte offsets (buffer_load ... lds): a geometry whose activation tensor
    would exceed 4 GiB is refused when it is built, not mis-addressed at run time."""
    from radet_amd import _lib
    from radet_amd.kernels import ConvGeom, Levels
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    ConvGeom(Levels([(120, 160)], 16), 256, 256, 1, 1, 0)                     # the reference config's largest: 314 MB
    with pytest.raises(_lib.RadetHipError, match="4 GiB"):
        ConvGeom(Levels([(800, 1344)], 16), 256, 64, 1, 1, 0)                 # 17.6 GB


def test_host_only_entry_points():
    """Pure host arithmetic exported by the library (no device access)."""
    from radet_amd import _lib
    lib = _lib.load()
    s = lib.radet_conv2d_wgrad_splits(25600, 256, 256, 3, 3)
    assert 1 <= s <= 64
    assert lib.radet_head_loss_ws_ints(25600) > 25600
    assert lib.radet_nms_ws_bytes(2, 5000) >= 2 * 5000 * 40
    assert lib.radet_assign_ws_bytes(4, 6400) >= 4 * 6400 * 32


def test_config_and_registry_dropin():
    from radet_amd import models
    from radet_amd.utils import Config
    cfg = Config.fromfile(CFG)
    assert cfg.model.type == "RADet" and cfg.model.backbone.depth == 50 and cfg.test_cfg.nms.type == "vote"
    assert cfg.optimizer.type == "AdamW" and cfg.optimizer_config.grad_clip.max_norm == 35
    cfg.merge_from_dict({"model.backbone.depth": 101})
    assert cfg.model.backbone.depth == 101
    for reg, names in ((models.BACKBONES, ["ResNet"]), (models.NECKS, ["FPN"]), (models.HEADS, ["RADetHead"]),
                       (models.LOSSES, ["FocalLoss", "GIoULoss", "CrossEntropyLoss"]), (models.DETECTORS, ["RADet"])):
        for n in names:
            assert n in reg
    from radet_amd.core import ANCHOR_GENERATORS, BBOX_ASSIGNERS, BBOX_CODERS, BBOX_SAMPLERS
    assert "AnchorGenerator" in ANCHOR_GENERATORS and "TBLRBBoxCoder" in BBOX_CODERS
    assert "MaxIoUAssigner" in BBOX_ASSIGNERS and "PseudoSampler" in BBOX_SAMPLERS
    from radet_amd.datasets import PIPELINES
    assert "LabelAssignment" in PIPELINES and "GenerateDistanceMap" in PIPELINES
    with pytest.raises(KeyError):
        models.build_backbone(dict(type="NoSuchNet"))


def test_reference_config_file_loads_unchanged(tmp_path, monkeypatch):
    """The reference's own config file builds the detector; its `pretrained='torchvision://resnet50'` is resolved
    offline (RADET_PRETRAINED_DIR) and loaded into the BACKBONE (single_stage.py:36-52, resnet.py:590-599); a missing
    file is an error, never a silent random init."""
    ref = "/root/reference/configs/bop/r50_ycbv_pbr.py"
    if not os.path.exists(ref):
        ref = CFG                      # same keys / values (tests/test_host.py::test_config_matches...), travels to the GPU box
    from radet_amd.models import build_detector
    from radet_amd.utils import Config
    cfg = Config.fromfile(ref)
    assert cfg.model["pretrained"] == "torchvision://resnet50"
    monkeypatch.setenv("RADET_PRETRAINED_DIR", str(tmp_path))
    monkeypatch.setenv("TORCH_HOME", str(tmp_path / "no_hub"))
    with pytest.raises(FileNotFoundError):
        build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    # a torchvision-format checkpoint: bare ResNet keys ('conv1.weight', 'layer1.0.bn1.running_mean', 'fc.weight', ...)
    cfg.model["pretrained"] = None
    donor = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    g = torch.Generator().manual_seed(9)
    tv = {k: torch.randn(v.shape, generator=g) if v.is_floating_point() else v.clone()
          for k, v in donor.backbone.state_dict().items()}
    tv["fc.weight"], tv["fc.bias"] = torch.zeros(1000, 2048), torch.zeros(1000)
    torch.save(tv, tmp_path / "resnet50.pth")
    cfg.model["pretrained"] = "torchvision://resnet50"
    det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    assert det.bbox_head.test_cfg.nms.iou_threshold == 0.65 and det.bbox_head.train_cfg.assigner.type == "MaxIoUAssigner"
    sd = det.backbone.state_dict()
    assert all(torch.equal(sd[k], tv[k]) for k in sd)
    assert not det.backbone.conv1.weight.requires_grad and det.backbone.layer2[0].conv1.weight.requires_grad
    # a detector checkpoint ('backbone.*' keys, mmcv 'state_dict' wrapper) also lands in the backbone
    torch.save(dict(state_dict={"backbone." + k: v * 2 for k, v in tv.items() if k.startswith(("conv1", "layer"))}),
               tmp_path / "det.pth")
    cfg.model["pretrained"] = str(tmp_path / "det.pth")
    det2 = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    assert torch.equal(det2.backbone.layer3[1].conv2.weight, tv["layer3.1.conv2.weight"] * 2)
    torch.save({"unrelated.weight": torch.zeros(3)}, tmp_path / "junk.pth")
    cfg.model["pretrained"] = str(tmp_path / "junk.pth")
    with pytest.raises(RuntimeError):
        build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)


def test_trainable_stem_in_bf16_storage_is_refused():
    """frozen_stages=-1 trains conv1 / bn1 (built for the fp32-tensor modes, tests/test_gpu_configs.py); with bf16 storage there
    is no stem backward, so it must not be accepted silently (zero gradients + weight decay would shrink the stem)."""
    from radet_amd.engine import Engine
    from radet_amd.models import build_detector
    from radet_amd.runtime import FlatParams
    from radet_amd.utils import Config
    cfg = Config.fromfile(CFG)
    cfg.model["pretrained"] = None
    cfg.model["backbone"]["frozen_stages"] = -1
    det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    flat = FlatParams(det, torch.device("cpu"))
    assert "backbone.conv1.weight" in flat.train_names and "backbone.bn1.bias" in flat.train_names
    with pytest.raises(NotImplementedError):
        Engine(flat.p, flat.g, depth=50, frozen_stages=-1, math="bf16-storage")


def build(depth=50):
    from radet_amd.models import build_detector
    from radet_amd.utils import Config
    cfg = Config.fromfile(CFG)
    cfg.model["pretrained"] = None
    cfg.model["backbone"]["depth"] = depth
    return build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)


@pytest.mark.parametrize("depth", [50, 101])
def test_state_dict_matches_reference_names(depth):
    from oracle import model as om
    det = build(depth)
    sd, ref = det.state_dict(), om.make_state_dict(depth)
    assert list(sd.keys()) == list(ref.keys())
    assert all(tuple(sd[k].shape) == tuple(ref[k].shape) for k in ref)
    train = {n for n, p in det.named_parameters() if p.requires_grad}
    assert train == {n for n in ref if ref[n].is_floating_point() and om.is_trainable(n)}
    if depth == 50:
        assert sum(p.numel() for p in det.parameters()) == 32159327
        assert sum(p.numel() for p in det.parameters() if p.requires_grad) == 31933983


def test_reference_init_statistics():
    det = build()
    sd = det.state_dict()
    assert float(sd["backbone.layer1.0.bn3.weight"].abs().sum()) == 0.0          # zero_init_residual
    assert float(sd["backbone.layer1.0.bn1.weight"].mean()) == 1.0
    assert abs(float(sd["bbox_head.atss_cls.bias"].mean()) + np.log(99.0)) < 1e-6  # bias_init_with_prob(0.01)
    assert abs(float(sd["bbox_head.cls_convs.0.conv.weight"].std()) - 0.01) < 1e-3
    assert float(sd["bbox_head.scales.3.scale"]) == 1.0
    w = sd["neck.lateral_convs.0.conv.weight"]
    bound = (6.0 / (w.shape[1] + w.shape[0])) ** 0.5                                # xavier uniform
    assert float(w.abs().max()) <= bound + 1e-6


def test_no_cpu_fallback():
    """The product path must fail loudly off-GPU instead of computing on the host."""
    from radet_amd._lib import RadetHipError
    det = build()
    img = torch.zeros(1, 3, 64, 64)
    with pytest.raises(RadetHipError):
        det.extract_feat(img)
    with pytest.raises(RadetHipError):
        det(img=img, img_metas=[dict(img_shape=(64, 64, 3))], return_loss=True, gt_bboxes=[torch.zeros(0, 4)],
            gt_labels=[torch.zeros(0, dtype=torch.long)], points_to_gt_index=[torch.zeros(85)], points_weight=[torch.ones(85)])


def test_geometry():
    from radet_amd.kernels import ConvGeom, Levels
    lv = Levels([(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)], 4)
    assert lv.rows == 4 * 6400 and lv.offsets == [0, 19200, 24000, 25200, 25520]
    g = ConvGeom(Levels([(15, 20)], 2), 256, 256, 3, 2, 1)
    assert g.lout.hw == [(8, 10)] and g.nseg == 1
    g = ConvGeom(Levels([(480 // 4, 640 // 4)], 4), 256, 512, 1, 2, 0)
    assert g.lout.hw == [(60, 80)]
    from radet_amd.engine import Engine
    sd = build().state_dict()
    e = Engine({k: v for k, v in sd.items()}, {}, depth=50)
    assert len(e.convs) == 53 + 8 + 11
    assert sum(c.wsize for c in e.convs) == sum(v.numel() for k, v in sd.items() if v.dim() == 4)
    assert [c.need_dgrad for c in e.convs[:5]] == [False] * 5           # stem + frozen layer1
    l2 = [c for c in e.convs if c.name.startswith("backbone.layer2.0.")]
    assert {c.name.split(".")[-1]: c.need_dgrad for c in l2} == {"conv1": False, "conv2": True, "conv3": True, "0": False}


def test_bbox2result_and_anchor_cpu_grid():
    from radet_amd.core import bbox2result, build_anchor_generator
    dets = torch.tensor([[0., 0., 1., 1., .9], [1., 1., 2., 2., .8], [2., 2., 3., 3., .7]])
    res = bbox2result(dets, torch.tensor([2, 0, 2]), 21)
    assert len(res) == 21 and res[2].shape == (2, 5) and res[0].shape == (1, 5) and res[1].shape == (0, 5)
    ag = build_anchor_generator(dict(type="AnchorGenerator", ratios=[1.0], octave_base_scale=8, scales_per_octave=1,
                                     strides=[8, 16, 32, 64, 128]))
    a = ag.grid_anchors([(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)], device="cpu")
    g = np.load(os.path.join(REPO, "tests", "golden", "anchors.npz"))
    assert np.array_equal(torch.cat(a).numpy(), g["a480x640"])


def test_reference_import_paths():
    """`radet.*` import paths of the reference resolve to the MI355X implementation."""
    from radet.models import DETECTORS, build_detector  # noqa: F401
    from radet.core import build_anchor_generator, bbox2result  # noqa: F401
    from radet.ops import vote_nms, global_vote_nms, cluster_nms  # noqa: F401
    from radet.datasets import PIPELINES
    import radet_amd.models
    assert DETECTORS is radet_amd.models.DETECTORS and "LabelAssignment" in PIPELINES
    # sub-modules resolve to the SAME module objects (a second import under the alias name would re-register everything)
    import radet.datasets.pipelines as p
    import radet_amd.datasets.pipelines as q
    from radet.core import bbox_overlaps, multiclass_nms  # noqa: F401
    from radet.core.bbox import TBLRBBoxCoder  # noqa: F401
    from radet.datasets import BOPDataset, build_dataset  # noqa: F401
    from radet.models.losses import FocalLoss, GIoULoss  # noqa: F401
    from radet.apis import train_detector  # noqa: F401
    assert p is q


def test_onecycle_matches_torch():
    """mmcv's OneCycle hook == torch.optim.lr_scheduler.OneCycleLR (linear, two-phase) for the config's values."""
    from radet_amd.apis import OneCycleLR
    total, max_lr, pct = 1000, 4e-4, 0.05
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=max_lr)
    ref = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=max_lr, total_steps=total, pct_start=pct,
                                              anneal_strategy="linear", div_factor=25.0, final_div_factor=1e4)
    mine = OneCycleLR(max_lr, total, pct_start=pct, anneal_strategy="linear")
    for step in range(total):
        assert abs(opt.param_groups[0]["lr"] - mine.get_lr(step)) <= 1e-12 + 1e-9 * max_lr, step
        opt.step()
        if step + 1 < total:
            ref.step()
    assert abs(mine.get_lr(0) - 1.6e-5) < 1e-12 and abs(mine.get_lr(total - 1) - 1.6e-9) < 1e-15


def test_checkpoint_roundtrip_cpu(tmp_path):
    from radet_amd.apis import load_checkpoint, save_checkpoint
    a, b = build(), build()
    for p in a.parameters():
        torch.nn.init.normal_(p, 0, 0.02)
    path = save_checkpoint(a, str(tmp_path / "iter_{iter}.pth").format(iter=7), meta=dict(iter=7, CLASSES=("a", "b")))
    meta, _ = load_checkpoint(b, path, strict=True)
    assert meta["iter"] == 7
    sa, sb = a.state_dict(), b.state_dict()
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
    # a reference-style checkpoint saved from a DDP-wrapped model ("module." prefix) also loads
    torch.save(dict(state_dict={"module." + k: v for k, v in sa.items()}, meta={}), str(tmp_path / "ddp.pth"))
    c = build()
    load_checkpoint(c, str(tmp_path / "ddp.pth"), strict=True)
    assert torch.equal(c.state_dict()["bbox_head.atss_cls.weight"], sa["bbox_head.atss_cls.weight"])


def test_generate_distance_map_crops_properties():
    """GenerateDistanceMap(with_gt_mask=False).crop_boxes (loading.py:596-634): canvas = box grown by ceil(pad_ratio * side),
    filled with ONE colour per box drawn as three `random.randint(0, 255)` in box order, overlaid with the image where the
    grown window is inside it; regions = the box inside its canvas; small boxes flagged."""
    import random
    from radet_amd.datasets.pipelines import GenerateDistanceMap
    g = GenerateDistanceMap.__new__(GenerateDistanceMap)
    g.with_gt_mask, g.small_object_size, g.pad_ratio = False, 32 ** 2, 0.05
    rng = np.random.RandomState(3)
    img = rng.randint(0, 256, (480, 640, 3)).astype(np.uint8)
    boxes = np.array([[100.7, 50.2, 300.9, 250.1],      # interior
                      [0.0, 0.0, 90.5, 70.5],            # grown window leaves the image at the top left
                      [560.2, 400.3, 639.9, 479.9],      # ... and at the bottom right
                      [10.0, 10.0, 30.0, 35.0]], np.float32)   # small object
    random.seed(11)
    crops, large, regions = g.crop_boxes(img, (480, 640), boxes)
    random.seed(11)
    colours = np.array([random.randint(0, 255) for _ in range(12)], np.uint8).reshape(4, 3)
    after = random.random()
    random.seed(11); [random.randint(0, 255) for _ in range(12)]
    assert random.random() == after                      # exactly 3 draws per box
    assert large.tolist() == [True, True, True, False]
    ib = boxes.astype(np.int_)
    for k, (x0, y0, x1, y1) in enumerate(ib):
        px, py = math.ceil((x1 - x0) * 0.05), math.ceil((y1 - y0) * 0.05)
        assert crops[k].shape == (y1 - y0 + 2 * py, x1 - x0 + 2 * px, 3) and crops[k].dtype == np.uint8
        assert regions[k].tolist() == [px, py, x1 - x0 + px, y1 - y0 + py]
        for cy in range(crops[k].shape[0]):               # every canvas pixel: image where the window covers it, else the colour
            iy = y0 - py + cy
            for cx in (0, px, crops[k].shape[1] // 2, crops[k].shape[1] - 1):
                ix = x0 - px + cx
                inside = 0 <= iy < min(479, y1 + py) and 0 <= ix < min(639, x1 + px)   # (the reference's clip to H-1 / W-1 is exclusive)
                want = img[iy, ix] if inside else colours[k]
                assert (crops[k][cy, cx] == want).all(), (k, cy, cx)


def test_label_assignment_refuses_anchor_generators_it_does_not_model():
    """adapt_positive_num reads the anchor side of a level (label_assignment.py:88-110); kernel and oracle use 8 * stride.
    Any anchor_generator_cfg that would give another size must raise instead of silently changing K per gt."""
    from radet_amd.datasets.pipelines import LabelAssignment
    std = dict(type="AnchorGenerator", ratios=[1.0], octave_base_scale=8, scales_per_octave=1, strides=[8, 16, 32, 64, 128])
    LabelAssignment(adapt_positive_num=True, anchor_generator_cfg=std)
    LabelAssignment(adapt_positive_num=True)
    for bad in (dict(std, ratios=[0.5, 1.0]), dict(std, octave_base_scale=4), dict(std, scales_per_octave=3),
                dict(std, strides=[8, 16, 32, 64, 256]), dict(std, center_offset=0.5)):
        with pytest.raises(NotImplementedError):
            LabelAssignment(adapt_positive_num=True, anchor_generator_cfg=bad)
        LabelAssignment(adapt_positive_num=False, anchor_generator_cfg=bad)      # unused without adapt_positive_num, as in the reference


def test_fold_kernel_float_reciprocal_quotients_are_exact():
    """radet_amd/csrc/layers.hip fold_kernel (round 6) takes its index quotients as (int)((i + 0.5f) * (1.0f / n)) instead of
    integer divisions: i < 32 * 288 (elements of one work item), n <= 288 (run-time extents of the item).  Exact over that
    whole range, also when the reciprocal is a few ulps off (a hardware rcp)."""
    i = np.arange(0, 32 * 288, dtype=np.int64)
    for n in range(1, 289):
        inv = np.float32(1.0) / np.float32(n)
        for d in (-2, 0, 2):
            invd = inv
            for _ in range(abs(d)):
                invd = np.nextafter(invd, np.float32(np.inf if d > 0 else -np.inf), dtype=np.float32)
            q = ((i.astype(np.float32) + np.float32(0.5)) * invd).astype(np.int64)
            assert np.array_equal(q, i // n), (n, d)
