"""GPU parity tests (pytest -m gpu): every HIP kernel is called through the C ABI and compared with
the oracle / golden vectors (ints bit-exact, fp32 within the stated tolerance)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-4   # north_star: fp32 boxes / scores within 1e-4


@pytest.fixture(scope="module")
def K():
    from radet_amd import kernels
    return kernels


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def to_rows(x):   # NCHW -> [B*H*W, C]
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


def from_rows(r, B, H, W):
    return r.reshape(B, H, W, -1).permute(0, 3, 1, 2)


def fold_w(w):    # OIHW -> [O][kh*kw][I]
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1, w.shape[1]).contiguous()


CONV_CASES = [
    # B, Cin, Cout, H, W, k, stride, tile
    (2, 64, 64, 20, 24, 1, 1, 0),
    (2, 64, 256, 20, 24, 1, 1, 0),
    (1, 128, 128, 17, 23, 3, 1, 0),
    (2, 128, 128, 18, 22, 3, 2, 0),
    (2, 256, 512, 15, 20, 1, 2, 0),
    (1, 256, 21, 15, 20, 3, 1, 0),
    (1, 256, 4, 9, 11, 3, 1, 0),
    (1, 256, 1, 9, 11, 3, 1, 0),
    (2, 256, 256, 30, 40, 3, 1, 1),
    (2, 256, 256, 30, 40, 3, 1, 2),
    (2, 256, 256, 30, 40, 3, 1, 3),
    (1, 512, 512, 15, 20, 3, 1, 0x3003),     # forced split-K = 3, 64x64 tiles
    (1, 1024, 256, 15, 20, 1, 1, 0x2000),    # forced split-K = 2, heuristic tile
    (4, 2048, 512, 15, 20, 1, 1, 0),         # layer4 shape: heuristic picks split-K
    (4, 256, 256, 80, 80, 3, 1, 0x203),      # 1600 tiles: the 64 left-over tiles are split along K (tail split)
    (3, 128, 256, 40, 56, 3, 1, 2),          # 53 x 4 = 212... tiles with a ragged last M tile
    (2, 256, 21, 48, 64, 3, 1, 0),           # predictor heads at M >= 4096: single-wave all-taps wgrad kernel
    (2, 256, 4, 48, 66, 3, 1, 0),
    (3, 256, 1, 40, 40, 3, 1, 0),
]


@pytest.mark.parametrize("math", [0, 1, 2, 3], ids=["fp32mfma", "bf16math", "fp32x3", "fp32h2"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(K, case, math):
    """math=1: operands rounded (RNE) to bf16 inside the kernel, fp32 accumulate -- the reference is the exact
    convolution of the pre-rounded tensors, so the tolerance stays at fp32-accumulation level.
    math=2 (the default fp32 arithmetic): fp32 operands split exactly into three bf16 planes, 6 of the 9 plane products
    on the bf16 matrix cores -- held to the SAME fp64 reference and tolerance as the native fp32 MFMA path (math=0).
    math=3 (the default fp32 arithmetic since round 5): fp32 operands scaled by a power of two and split into two fp16 planes,
    3 plane products on the f16 matrix cores (the operands' amax slots are computed on the spot here) -- same reference and
    tolerance again."""
    B, Cin, Cout, H, W, k, s, tile = case
    x3, math = {2: True, 3: "h2"}.get(math, False), math & 1 if math < 2 else 0
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout, generator=g)
    pad = k // 2
    r = (lambda t: t.bfloat16().double()) if math else (lambda t: t.double())
    y_ref = F.conv2d(r(x), r(w), bias.double(), stride=s, padding=pad)
    Ho, Wo = y_ref.shape[2:]
    res = torch.randn(B, Cout, Ho, Wo, generator=g)
    out_ref = F.relu(y_ref + res.double())
    dy = torch.randn(B, Cout, Ho, Wo, generator=g)
    gx = torch.nn.grad.conv2d_input(x.shape, r(w), r(dy), stride=s, padding=pad)
    gw = torch.nn.grad.conv2d_weight(r(x), w.shape, r(dy), stride=s, padding=pad)

    dev = "cuda"
    lv = K.Levels([(H, W)], B)
    geom = K.ConvGeom(lv, Cin, Cout, k, s, pad)
    geom.math = math
    geom.x3 = x3
    xr, wf = to_rows(x).to(dev), fold_w(w).to(dev)
    y = torch.empty(B * Ho * Wo, Cout, device=dev)
    K.conv_fwd(geom, xr, wf, bias.to(dev), y, addend=to_rows(res).to(dev), relu=True, tile=tile)
    assert rel_err(from_rows(y, B, Ho, Wo), out_ref) < 1e-5
    # dgrad (needs K = Cout multiple of 16 -> pad like the engine does for the small predictors)
    kc = (Cout + 15) // 16 * 16
    dyr = torch.zeros(B * Ho * Wo, kc, device=dev)
    dyr[:, :Cout] = to_rows(dy).to(dev)
    wft = torch.zeros(Cin, k * k, kc, device=dev)
    wft[:, :, :Cout] = w.permute(1, 2, 3, 0).reshape(Cin, k * k, Cout).to(dev)
    dx = torch.empty(B * H * W, Cin, device=dev)
    mask = to_rows(torch.randn(B, Cin, H, W, generator=g)).to(dev)
    K.conv_dgrad(geom, dyr, wft.contiguous(), dx, mask=mask, k_channels=kc, tile=tile)
    gx_m = gx * (from_rows(mask.cpu(), B, H, W) > 0)
    assert rel_err(from_rows(dx, B, H, W), gx_m) < 1e-5
    # wgrad + fused bias partials
    S = geom.nsplit
    slabs = torch.empty(S, Cout, k * k, Cin, device=dev)
    bp = torch.empty(S, Cout, device=dev)
    K.conv_wgrad(geom, dyr, xr, slabs, bp, cout=Cout, ld_dy=kc)
    gw_mine = slabs.sum(0).reshape(Cout, k, k, Cin).permute(0, 3, 1, 2)
    assert rel_err(gw_mine, gw) < 2e-5
    assert rel_err(bp.sum(0), dy.double().sum((0, 2, 3))) < 2e-5


KDIV_CASES = [
    # B, Cin, Cout, H, W, k, stride, split-K
    (2, 256, 256, 30, 40, 3, 1, 0),          # layer3 3x3
    (1, 512, 512, 15, 20, 3, 1, 3),          # layer4 3x3, forced split-K = 3
    (2, 1024, 256, 15, 20, 1, 1, 2),         # 1x1 down, split-K = 2
    (3, 128, 192, 17, 23, 3, 1, 0),          # ragged M tile, ragged N tile (192 = 3 x 64), K = 128
    (2, 128, 128, 18, 22, 3, 2, 0),          # strided: the dgrad is a class launch
    (2, 256, 512, 15, 20, 1, 2, 0),          # 1x1 / 2: dgrad classes with rows that receive no tap
    (1, 64, 64, 9, 11, 1, 1, 0),             # a single K stage
]


@pytest.mark.parametrize("arith", [True, "h2"], ids=["bf16x3", "fp16x2"])
@pytest.mark.parametrize("kd", [7, 8, 8 | 0x20000, 8 | 0x40000], ids=["4-kgroups", "2-kgroups", "2-kgroups-3stages", "2-kgroups-4stages"])
@pytest.mark.parametrize("case", KDIV_CASES)
def test_k_divided_tiles(K, case, kd, arith):
    """The 64 x 64 tiles whose four waves divide the K step (tile 7: four 16-channel k-groups of a 64-channel stage, every
    wave accumulates the whole tile; tile 8: two k-groups x two column halves of a 32-channel stage) and add their partial
    tiles through LDS; and the pixel-divided one-tap wgrad (flags 0x400 / 0x800).  Same fp64 reference and tolerance as
    every other fp32 path; bit-identical from run to run."""
    B, Cin, Cout, H, W, k, s, sk = case
    if kd > 8 and arith != "h2":
        pytest.skip("deeper pipelines of the K-divided tile exist for the fp16 hi / lo arithmetic only")
    g = torch.Generator().manual_seed(sum(case) + (kd & 0xFF) + (kd >> 17))
    kd, deep = kd & 0xFF, kd & ~0xFF
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout, generator=g)
    pad = k // 2
    y_ref = F.conv2d(x.double(), w.double(), bias.double(), stride=s, padding=pad)
    Ho, Wo = y_ref.shape[2:]
    res = torch.randn(B, Cout, Ho, Wo, generator=g)
    out_ref = F.relu(y_ref + res.double())
    dy = torch.randn(B, Cout, Ho, Wo, generator=g)
    gx = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), stride=s, padding=pad)
    gw = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), stride=s, padding=pad)
    dev = "cuda"
    lv = K.Levels([(H, W)], B)
    geom = K.ConvGeom(lv, Cin, Cout, k, s, pad)
    geom.x3 = arith
    tile = kd | deep | (sk << 12)
    xr, wf = to_rows(x).to(dev), fold_w(w).to(dev)
    y = torch.full((B * Ho * Wo, Cout), float("nan"), device=dev)
    y2 = torch.full_like(y, float("nan"))
    K.conv_fwd(geom, xr, wf, bias.to(dev), y, addend=to_rows(res).to(dev), relu=True, tile=tile)
    K.conv_fwd(geom, xr, wf, bias.to(dev), y2, addend=to_rows(res).to(dev), relu=True, tile=tile)
    assert torch.equal(y, y2)
    assert rel_err(from_rows(y, B, Ho, Wo), out_ref) < 1e-5
    if Cout % (64 if kd == 7 else 32) == 0:
        dyr = to_rows(dy).to(dev)
        wft = w.permute(1, 2, 3, 0).reshape(Cin, k * k, Cout).contiguous().to(dev)
        dx = torch.full((B * H * W, Cin), float("nan"), device=dev)
        mask = to_rows(torch.randn(B, Cin, H, W, generator=g)).to(dev)
        K.conv_dgrad(geom, dyr, wft, dx, mask=mask, tile=tile)
        gx_m = gx * (from_rows(mask.cpu(), B, H, W) > 0)
        assert rel_err(from_rows(dx, B, H, W), gx_m) < 1e-5
    # pixel-divided wgrad, a few split counts (the last split is ragged)
    dyr = to_rows(dy).to(dev)
    for S in (1, 3):
        geom.wgrad_flags, geom.nsplit = (2 << 4) | 0x40 | (0x400 if kd == 7 else 0x800), S
        slabs = torch.full((S, Cout, k * k, Cin), float("nan"), device=dev)
        bp = torch.full((S, Cout), float("nan"), device=dev)
        K.conv_wgrad(geom, dyr, xr, slabs, bp)
        gw_mine = slabs.sum(0).reshape(Cout, k, k, Cin).permute(0, 3, 1, 2)
        assert rel_err(gw_mine, gw) < 2e-5, S
        assert rel_err(bp.sum(0), dy.double().sum((0, 2, 3))) < 2e-5, S


@pytest.mark.parametrize("x3", [False, True, "h2"], ids=["fp32mfma", "fp32x3", "fp32h2"])
@pytest.mark.parametrize("shape", [
    # B, Cin, Cout, H, W, tile
    (2, 128, 128, 18, 22, 0),
    (1, 256, 256, 13, 7, 0x5203),       # odd sizes (classes of unequal size, ragged last tiles), forced split-K 5
    (4, 256, 256, 15, 20, 0x3203),      # the FPN P6 shape
    (2, 64, 128, 31, 33, 2),
    (3, 512, 512, 9, 11, 0x201),        # 128 x 128 tiles over classes shorter than one tile
])
def test_strided_dgrad_class_launch(K, shape, x3, monkeypatch):
    """All four parity classes of a 3x3 / 2 dgrad in one launch (radet_conv2d_igemm_classes) against the fp64 dgrad, with
    addend + mask, and against the one-launch-per-class path it replaces."""
    B, Cin, Cout, H, W, tile = shape
    g = torch.Generator().manual_seed(sum(shape))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dy = torch.randn(B, Cout, Ho, Wo, generator=g)
    add = torch.randn(B, Cin, H, W, generator=g)
    msk = torch.randn(B, Cin, H, W, generator=g)
    gx = torch.nn.grad.conv2d_input((B, Cin, H, W), w.double(), dy.double(), stride=2, padding=1)
    ref = (gx + add.double()) * (msk > 0)
    geom = K.ConvGeom(K.Levels([(H, W)], B), Cin, Cout, 3, 2, 1)
    geom.x3 = x3
    assert K._strided_dgrad_group(geom) is not None and K._strided_dgrad_group(geom)["ncls"] == 4
    dev = "cuda"
    dyr = to_rows(dy).to(dev)
    wft = w.permute(1, 2, 3, 0).reshape(Cin, 9, Cout).contiguous().to(dev)
    addr, mr = to_rows(add).to(dev), to_rows(msk).to(dev)
    dx = torch.full((B * H * W, Cin), float("nan"), device=dev)
    K.conv_dgrad(geom, dyr, wft, dx, addend=addr, mask=mr, tile=tile)
    assert rel_err(from_rows(dx, B, H, W), ref) < 1e-5
    geom2 = K.ConvGeom(K.Levels([(H, W)], B), Cin, Cout, 3, 2, 1)
    geom2.x3 = x3
    monkeypatch.setattr(K, "STRIDED_DGRAD_GROUP", False)
    assert K._strided_dgrad_group(geom2) is None
    dx2 = torch.full((B * H * W, Cin), float("nan"), device=dev)
    K.conv_dgrad(geom2, dyr, wft, dx2, addend=addr, mask=mr, tile=tile)
    assert rel_err(dx2, dx.double()) < 2e-6              # (split-K factors may differ between the two paths)
    # in place on a pre-masked dx (the engine's residual accumulation)
    monkeypatch.setattr(K, "STRIDED_DGRAD_GROUP", True)
    dx3 = (addr * (mr > 0)).contiguous()
    K.conv_dgrad(geom, dyr, wft, dx3, addend=dx3, mask=mr, tile=tile, skip_zero_rows=True)
    assert rel_err(from_rows(dx3, B, H, W), ref) < 1e-5


@pytest.mark.parametrize("hw,B", [([(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)], 2), ([(13, 7), (9, 33)], 3),
                                  ([(100, 100), (50, 50), (25, 25), (13, 13), (7, 7)], 1)])
def test_predictor_convs_from_lds_patch(K, hw, B):
    """radet_pred3x3_patch (cls alone; reg + iou sharing one launch) against the fp64 convolution per level and against
    the implicit-GEMM path it replaces, on ragged multi-level pyramids."""
    Cin = 256
    g = torch.Generator().manual_seed(len(hw) * 100 + B)
    lv = K.Levels(hw, B)
    xs = [torch.randn(B, Cin, h, w, generator=g) for h, w in hw]
    x = torch.cat([to_rows(t) for t in xs]).cuda()
    heads = {}
    for name, c in (("cls", 21), ("reg", 4), ("iou", 1)):
        w = torch.randn(c, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
        heads[name] = (w, torch.randn(c, generator=g), torch.full((lv.rows, c), float("nan"), device="cuda"), c)
    q = lambda n: (fold_w(heads[n][0]).cuda(), heads[n][1].cuda(), heads[n][2], heads[n][3])  # noqa: E731
    K.pred_conv_patch(lv, x, q("cls"))
    K.pred_conv_patch(lv, x, q("reg"), q("iou"))
    geom = K.ConvGeom(lv, Cin, 21, 3, 1, 1)
    geom.x3 = True
    y_ig = torch.empty(lv.rows, 21, device="cuda")
    K.conv_fwd(geom, x, fold_w(heads["cls"][0]).cuda(), heads["cls"][1].cuda(), y_ig)
    for name, (w, b, y, c) in heads.items():
        ref = torch.cat([to_rows(F.conv2d(t.double(), w.double(), b.double(), padding=1)) for t in xs])
        assert torch.isfinite(y).all()
        assert rel_err(y.cpu(), ref) < 1e-5, name
    assert rel_err(heads["cls"][2], y_ig.double()) < 2e-6


def test_fp32_from_bf16_planes_is_as_accurate_as_the_fp32_mfma(K):
    """The default fp32 arithmetic forms products from three bf16 planes per operand (exact split, 6 of 9 plane products,
    fp32 accumulate).  On the tower shape its error against an fp64 convolution must not exceed that of the native
    v_mfma_f32_32x32x2_f32 path (measured: 0.8x) -- forward, dgrad and wgrad; exact under power-of-two scalings like
    the native path; bit-identical from run to run."""
    B, C, H, W = 2, 256, 40, 40
    gen = torch.Generator().manual_seed(9)
    x4 = torch.randn(B, C, H, W, generator=gen)
    w4 = torch.randn(C, C, 3, 3, generator=gen) / (C * 9) ** 0.5
    dy4 = torch.randn(B, C, H, W, generator=gen)
    ref_y = F.conv2d(x4.double(), w4.double(), padding=1)
    ref_gw = torch.nn.grad.conv2d_weight(x4.double(), w4.shape, dy4.double(), padding=1)
    lv = K.Levels([(H, W)], B)
    xr, dyr, wf = to_rows(x4).cuda(), to_rows(dy4).cuda(), fold_w(w4).cuda()
    errs = {}
    for x3 in (False, True):
        g = K.ConvGeom(lv, C, C, 3, 1, 1)
        g.x3 = x3
        y = torch.empty(lv.rows, C, device="cuda")
        K.conv_fwd(g, xr, wf, None, y, tile=0x202)
        y2 = torch.empty_like(y)
        K.conv_fwd(g, xr * 4, wf * 0.5, None, y2, tile=0x202)
        assert torch.equal(y2, y * 2)
        slabs = torch.empty(g.nsplit, C, 9, C, device="cuda")
        K.conv_wgrad(g, dyr, xr, slabs)
        s2 = torch.empty_like(slabs)
        K.conv_wgrad(g, dyr, xr, s2)
        assert torch.equal(slabs, s2)
        gw = slabs.double().sum(0).reshape(C, 3, 3, C).permute(0, 3, 1, 2)
        errs[x3] = (rel_err(from_rows(y, B, H, W), ref_y), rel_err(gw, ref_gw.cuda() if gw.is_cuda else ref_gw))
        if x3:      # every one-tap tile of the plane arithmetic, 16- and 32-pixel stages (0x40: not the all-taps kernel)
            for fl in (0x40 | (1 << 4), 0x40 | (2 << 4), 0x40 | (3 << 4), 0xC0 | (2 << 4), 0xC0 | (3 << 4)):
                g.wgrad_flags, g.nsplit = fl, 6
                s3 = torch.empty(6, C, 9, C, device="cuda")
                K.conv_wgrad(g, dyr, xr, s3)
                assert rel_err(s3.double().sum(0).reshape(C, 3, 3, C).permute(0, 3, 1, 2), ref_gw) < 1e-5, hex(fl)
    for native, planes in zip(errs[False], errs[True]):
        assert planes <= 1.25 * native + 1e-9, errs
        assert planes < 1e-5


def _conv_errs(K, x4, w4, dy4, mode, tile=0x202, wflags=None):
    """(forward, dgrad, wgrad) error of one 3x3 conv against fp64, relative to the largest reference magnitude"""
    B, C, H, W = x4.shape
    Co = w4.shape[0]
    ref_y = F.conv2d(x4.double(), w4.double(), padding=1)
    ref_gx = torch.nn.grad.conv2d_input(x4.shape, w4.double(), dy4.double(), padding=1)
    ref_gw = torch.nn.grad.conv2d_weight(x4.double(), w4.shape, dy4.double(), padding=1)
    lv = K.Levels([(H, W)], B)
    g = K.ConvGeom(lv, C, Co, 3, 1, 1)
    g.x3 = mode
    xr, dyr, wf = to_rows(x4).cuda(), to_rows(dy4).cuda(), fold_w(w4).cuda()
    wft = w4.permute(1, 2, 3, 0).reshape(C, 9, Co).contiguous().cuda()
    y, dx = torch.empty(lv.rows, Co, device="cuda"), torch.empty(lv.rows, C, device="cuda")
    K.conv_fwd(g, xr, wf, None, y, tile=tile)
    K.conv_dgrad(g, dyr, wft, dx, tile=tile)
    if wflags is not None:
        g.wgrad_flags, g.nsplit = wflags, 4
    slabs = torch.empty(g.nsplit, Co, 9, C, device="cuda")
    K.conv_wgrad(g, dyr, xr, slabs)
    gw = slabs.double().sum(0).reshape(Co, 3, 3, C).permute(0, 3, 1, 2)
    return (rel_err(from_rows(y, B, H, W), ref_y), rel_err(from_rows(dx, B, H, W), ref_gx), rel_err(gw, ref_gw)), (y, dx, slabs)


def test_fp32_from_fp16_pairs_is_as_accurate_as_the_fp32_mfma(K):
    """The default fp32 arithmetic (round 5): every operand is scaled by an exact power of two (from its amax slot) and split
    into two fp16 numbers hi + 2^-11 lo; hi hi' + 2^-11 (hi lo' + lo hi') is accumulated in fp32 by three f16 MFMAs per
    K = 16 step.  Acceptance gate: error against an fp64 convolution <= 1.25 x the native v_mfma_f32_32x32x2_f32 path's and
    < 1e-5 -- forward, dgrad and wgrad, every tile family -- on the tower shape AND on operands whose magnitudes span 2^24
    inside one tensor, gradients scaled by 2^-20, rows of zeros; exact under power-of-two scalings (the scale follows the
    operand); bit-identical from run to run."""
    B, C, H, W = 2, 256, 40, 40
    gen = torch.Generator().manual_seed(9)

    def lognormal(shape, lo, hi):
        return torch.randn(shape, generator=gen) * torch.exp2(torch.empty(shape).uniform_(lo, hi, generator=gen))

    cases = {
        "gaussian": (torch.randn(B, C, H, W, generator=gen), torch.randn(C, C, 3, 3, generator=gen) / (C * 9) ** 0.5,
                     torch.randn(B, C, H, W, generator=gen)),
        "relu activations": (torch.randn(B, C, H, W, generator=gen).clamp_min(0), torch.randn(C, C, 3, 3, generator=gen) * 0.02,
                             torch.randn(B, C, H, W, generator=gen)),
        "log-uniform 2^+-12": (lognormal((B, C, H, W), -12, 12), lognormal((C, C, 3, 3), -12, 12) / (C * 9) ** 0.5,
                               lognormal((B, C, H, W), -12, 12)),
        "dy 2^-20": (torch.randn(B, C, H, W, generator=gen), torch.randn(C, C, 3, 3, generator=gen) / (C * 9) ** 0.5,
                     torch.randn(B, C, H, W, generator=gen) * 2.0 ** -20),
    }
    zr = torch.randn(B, C, H, W, generator=gen)
    zr[:, :, ::3] = 0                                          # rows of zeros (and whole zero channels of dy)
    zd = torch.randn(B, C, H, W, generator=gen)
    zd[:, ::2] = 0
    cases["zero rows"] = (zr, torch.randn(C, C, 3, 3, generator=gen) / (C * 9) ** 0.5, zd)
    for name, (x4, w4, dy4) in cases.items():
        for tile, wfl in ((0x202, 0x40 | (1 << 4)), (1, 0x40 | (3 << 4)), (3, 0xC0 | (2 << 4)), (7, 0x40 | (2 << 4) | 0x400),
                          (8, 0x40 | (2 << 4) | 0x800)):
            # the native instruction on the same tile and the same pixel splits (the summation structure is part of the error;
            # the K-divided tiles and the 128 x 64 weight-gradient tile exist for the plane arithmetics only: 64 x 64 there)
            native, _ = _conv_errs(K, x4, w4, dy4, False, tile=tile if (tile & 0xFF) < 7 else 3,
                                   wflags=(wfl & ~0xC00) if ((wfl >> 4) & 3) != 3 else 0x40 | (2 << 4))
            errs, outs = _conv_errs(K, x4, w4, dy4, "h2", tile=tile, wflags=wfl)
            for e_n, e_h, what in zip(native, errs, ("fwd", "dgrad", "wgrad")):
                assert e_h <= 1.25 * e_n + 1e-9, (name, hex(tile), what, e_h, e_n)
                assert e_h < 1e-5, (name, hex(tile), what, e_h)
            _, outs2 = _conv_errs(K, x4, w4, dy4, "h2", tile=tile, wflags=wfl)
            assert all(torch.equal(a, b) for a, b in zip(outs, outs2)), (name, hex(tile))      # run-to-run bit identity
        # power-of-two linearity: the operands' scales follow them exactly
        _, (y1, dx1, s1) = _conv_errs(K, x4, w4, dy4, "h2")
        _, (y2, dx2, s2) = _conv_errs(K, x4 * 4, w4 * 0.5, dy4 * 2.0 ** -7, "h2")
        assert torch.equal(y2, y1 * 2) and torch.equal(dx2, dx1 * 2.0 ** -8) and torch.equal(s2, s1 * 2.0 ** -5), name
    # an all-zero operand (amax slot 0: no scaling) and a tensor of one huge / one tiny magnitude
    x0 = torch.zeros(1, 64, 8, 8)
    w0 = torch.randn(64, 64, 3, 3, generator=gen)
    (e, _, _), (y, _, _) = _conv_errs(K, x0, w0, torch.zeros(1, 64, 8, 8), "h2", tile=3)
    assert float(y.abs().max()) == 0.0
    for mag in (2.0 ** 100, 2.0 ** -100):
        errs, _ = _conv_errs(K, torch.randn(1, 64, 8, 8, generator=gen) * mag, w0, torch.randn(1, 64, 8, 8, generator=gen) / mag,
                             "h2", tile=3)
        assert max(errs) < 1e-5, (mag, errs)


PAIR_CASES = [
    # B, Cin, Cout, H, W, k, tile
    (2, 256, 256, 30, 40, 3, 5),            # 128 x 128, 8 waves
    (2, 256, 256, 30, 40, 3, 6),            # 256 x 128, 8 waves
    (1, 128, 128, 17, 23, 3, 5),            # ragged last M tile, image-border taps
    (1, 512, 256, 15, 20, 3, 0x3005),       # forced split-K = 3
    (4, 256, 256, 80, 80, 3, 0x20006),      # 3 LDS stages, tail split
    (1, 1024, 256, 15, 20, 1, 0x2005),      # 1 x 1, forced split-K = 2
    (2, 64, 128, 20, 24, 1, 5),
]


@pytest.mark.parametrize("case", PAIR_CASES)
def test_plane_pair_operand_conv(K, case):
    """radet_conv2d_igemm_s with x / w as fp16 plane pairs (tile_override 0x2000000 | 0x8000000): the pair split represents
    every element to 2^-22 relative, and the conv is held to the same fp64 reference and tolerance as the fp32 paths (forward
    with bias + residual + ReLU, dgrad with mask); 3x3 cases also run the plane-pair all-taps wgrad (conv_wgrad9q_kernel)
    incl. its bias column sums."""
    B, Cin, Cout, H, W, k, tile = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout, generator=g)
    pad = k // 2
    y_ref = F.conv2d(x.double(), w.double(), bias.double(), padding=pad)
    res = torch.randn(B, Cout, H, W, generator=g)
    out_ref = F.relu(y_ref + res.double())
    dy = torch.randn(B, Cout, H, W, generator=g) * 1e-3
    gx = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), padding=pad)
    dev = "cuda"
    lv = K.Levels([(H, W)], B)
    geom = K.ConvGeom(lv, Cin, Cout, k, 1, pad)
    xr = to_rows(x).to(dev)
    xp = K.Planes.from_float(xr, kind="h2")
    assert xp.t.dtype == torch.float16 and xp.t.shape == (lv.rows, 2 * Cin)
    assert K.amax_value(xp.amax) == float(xr.abs().max())               # the slot holds the bit pattern of the largest |x|
    back = xp.to_float()
    assert float(((back - xr).abs() / xr.abs().clamp_min(float(xr.abs().max()) * 2.0 ** -26)).max()) <= 2.0 ** -22
    wp = K.Planes.from_float(fold_w(w).reshape(Cout * k * k, Cin).to(dev), kind="h2")
    y = torch.empty(lv.rows, Cout, device=dev)
    K.conv_fwd(geom, xp, wp, bias.to(dev), y, addend=to_rows(res).to(dev), relu=True, tile=tile)
    assert rel_err(from_rows(y, B, H, W), out_ref) < 1e-5
    dyr = to_rows(dy).to(dev)
    dyp = K.Planes.from_float(dyr, kind="h2")
    wtp = K.Planes.from_float(w.permute(1, 2, 3, 0).reshape(Cin * k * k, Cout).contiguous().to(dev), kind="h2")
    dx = torch.empty(lv.rows, Cin, device=dev)
    mask = to_rows(torch.randn(B, Cin, H, W, generator=g)).to(dev)
    K.conv_dgrad(geom, dyp, wtp, dx, mask=mask, tile=tile)
    gx_m = gx * (from_rows(mask.cpu(), B, H, W) > 0)
    assert rel_err(from_rows(dx, B, H, W), gx_m) < 1e-5
    if k == 3:
        gw = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), padding=pad)
        for S in (1, 5):                                      # (the last split is ragged)
            geom.nsplit = S
            slabs = torch.full((S, Cout, 9, Cin), float("nan"), device=dev)
            bp = torch.full((S, Cout), float("nan"), device=dev)
            K.conv_wgrad(geom, dyp, xp, slabs, bp)
            gw_mine = slabs.sum(0).reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
            assert rel_err(gw_mine, gw) < 2e-5, S
            assert rel_err(bp.sum(0), dy.double().sum((0, 2, 3))) < 2e-5, S


def test_plane_pair_outputs_of_groupnorm_and_fold(K):
    """GroupNorm + ReLU forward / backward with plane-PAIR outputs (radet_gn_relu_fwd_q / _bwd_q): the pairs reproduce the fp32
    outputs to 2^-22 of every element (2^-24 of the bound for the smallest ones), the bound written to the amax slot really
    bounds the tensor, the tracked slots (fp32 output, |zhat|) hold the exact maxima, and the statistics / parameter
    gradients are those of the fp32 kernels bit for bit."""
    dev = "cuda"
    lv = K.Levels([(12, 16), (6, 8), (3, 4)], 2)
    R = lv.rows
    g = torch.Generator().manual_seed(5)
    z = (torch.randn(R, 256, generator=g) * 3 + 0.5).to(dev)
    gam, bet = (torch.rand(256, generator=g) + 0.5).to(dev), (torch.randn(256, generator=g) * 0.1).to(dev)
    stats, stats2 = torch.empty(len(lv) * lv.B * 64, device=dev), torch.empty(len(lv) * lv.B * 64, device=dev)
    ws = torch.empty(K.gn_ws_floats(lv), device=dev)
    y = torch.empty_like(z)
    K.gn_relu_fwd(lv, z, gam, bet, y, stats, ws)

    def close(planes, ref):
        got = planes.to_float()
        bound = K.amax_value(planes.amax)
        assert bound >= float(ref.abs().max()), (bound, float(ref.abs().max()))
        tol = ref.abs() * 2.0 ** -22 + bound * 2.0 ** -36
        assert bool(((got - ref).abs() <= tol).all()), float(((got - ref).abs() / tol).max())

    yq, y2 = K.Planes(R, 256, device=dev, kind="h2"), torch.empty_like(z)
    ya, zh = K.new_amax(dev), K.new_amax(dev).fill_(77)
    key = K.register_amax(y2, ya)
    K.gn_relu_fwd_q(lv, z, gam, bet, y2, yq, stats2, ws, zhat_amax=zh)
    K.unregister_amax([key])
    assert torch.equal(y2, y) and torch.equal(stats2, stats)
    close(yq, y)
    assert K.amax_value(ya) == float(y.abs().max())
    zhat = torch.cat([((z[r0:r1].view(lv.B, -1, 32, 8) - m[:, None, :, None]) * r[:, None, :, None]).abs().reshape(-1)
                      for (r0, r1), m, r in ((lv.level_rows(l), stats.view(-1, 32, 2)[l * lv.B:(l + 1) * lv.B, :, 0],
                                              stats.view(-1, 32, 2)[l * lv.B:(l + 1) * lv.B, :, 1]) for l in range(len(lv)))])
    assert abs(K.amax_value(zh) - float(zhat.max())) <= 1e-5 * float(zhat.max())
    # pair launch
    zb = torch.randn(R, 256, generator=g).to(dev)
    yb, yqb = torch.empty_like(z), K.Planes(R, 256, device=dev, kind="h2")
    stats_b, ws_b = torch.empty_like(stats), torch.empty_like(ws)
    K.gn_relu_fwd_pair_q(lv, (z, gam, bet, None, yq, stats, ws, zh), (zb, gam, bet, yb, yqb, stats_b, ws_b, None))
    K.gn_relu_fwd(lv, zb, gam, bet, y2, stats_b, ws_b)
    close(yq, y)
    close(yqb, y2)
    assert torch.equal(yb, y2)
    # backward
    dy = (torch.randn(R, 256, generator=g) * 1e-4).to(dev)
    dz, dg, db = torch.empty_like(z), torch.empty(256, device=dev), torch.empty(256, device=dev)
    K.gn_relu_bwd(lv, dy, z, stats, gam, bet, dz, dg, db, ws)
    for zslot in (zh, None):                                   # tracked |zhat| bound, and the hard sqrt(n - 1) one
        dzq, dg2, db2 = K.Planes(R, 256, device=dev, kind="h2"), torch.empty(256, device=dev), torch.empty(256, device=dev)
        K.gn_relu_bwd_q(lv, dy, z, stats, gam, bet, None, dzq, dg2, db2, ws, zhat_amax=zslot)
        close(dzq, dz)
        assert torch.equal(dg2, dg) and torch.equal(db2, db)
    # folded weights as plane pairs (RadetConvDesc.w16 = 3) against the fp32 fold, through the engine's descriptor table
    from radet_amd import _lib
    Co, Ci = 64, 96
    w = torch.randn(Co, Ci, 3, 3, generator=g).to(dev)
    bn = [t.to(dev) for t in (torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g), torch.randn(Co, generator=g),
                              torch.rand(Co, generator=g) + 0.5)]
    wf32, wft32, bias_f = torch.empty(Co * 9 * Ci, device=dev), torch.empty(Ci * 9 * Co, device=dev), torch.empty(Co, device=dev)
    slots = K.new_amax(dev, 2)
    wfq = K.Planes(Co * 9, Ci, device=dev, kind="h2", amax=slots[1])
    wftq = K.Planes(Ci * 9, Co, device=dev, kind="h2", amax=slots[1])
    arr = (_lib.RadetConvDesc * 2)()
    for d, (a, b, w16, sl) in zip(arr, ((wf32, wft32, 0, slots[0]), (wfq.t, wftq.t, 3, slots[1]))):
        d.w, d.bn_gamma, d.bn_beta, d.bn_mean, d.bn_var = (C.c_void_p(t.data_ptr()) for t in (w, *bn))
        d.wf, d.wft, d.bias_f = C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(bias_f.data_ptr())
        d.cout, d.cin, d.kh, d.kw, d.eps, d.nsplit, d.w16 = Co, Ci, 3, 3, 1e-5, 1, w16
        d.w_amax = C.c_void_p(sl.data_ptr())
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    K.fold_weights(table, 2)
    assert K.amax_value(slots[0]) == K.amax_value(slots[1]) == float(wf32.abs().max())
    close(wfq, wf32.view(Co * 9, Ci))
    close(wftq, wft32.view(Ci * 9, Co))


@pytest.mark.parametrize("tile", [64, 128])
@pytest.mark.parametrize("math", [0, 1], ids=["fp32", "bf16math"])
def test_wgrad_group_launch(K, tile, math):
    """radet_conv2d_wgrad_group: several convs' weight gradients in one grid == exact convolution gradients, and
    bit-identical to the per-conv launches of the same kernel (same tile, same pixel splits)."""
    cases = [(2, 256, 128, 15, 20, 1, 1), (2, 128, 128, 18, 22, 3, 2), (1, 128, 256, 30, 40, 3, 1), (3, 512, 128, 8, 10, 1, 1),
             (2, 128, 128, 4, 5, 3, 2)]
    g = torch.Generator().manual_seed(11)
    jobs, refs, singles = [], [], []
    for (B, Cin, Cout, H, W, k, s) in cases:
        x = torch.randn(B, Cin, H, W, generator=g)
        pad = k // 2
        Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        dy = torch.randn(B, Cout, Ho, Wo, generator=g)
        r = (lambda t: t.bfloat16().double()) if math else (lambda t: t.double())
        refs.append((torch.nn.grad.conv2d_weight(r(x), (Cout, Cin, k, k), r(dy), stride=s, padding=pad), dy.double().sum((0, 2, 3))))
        geom = K.ConvGeom(K.Levels([(H, W)], B), Cin, Cout, k, s, pad)
        geom.math = math
        geom.nsplit = max(1, min(3, (B * Ho * Wo) // 256))
        S = geom.nsplit
        xr, dyr = to_rows(x).cuda(), to_rows(dy).cuda()
        slabs, bp = torch.zeros(S, Cout, k * k, Cin, device="cuda"), torch.zeros(S, Cout, device="cuda")
        jobs.append(dict(g=geom, dy=dyr, x=xr, slabs=slabs, dbias=bp))
        s1, b1 = torch.zeros_like(slabs), torch.zeros_like(bp)
        geom.wgrad_flags = ((1 if tile == 128 else 2) << 4) | 0x40
        K.conv_wgrad(geom, dyr, xr, s1, b1)
        singles.append((s1, b1))
    K.conv_wgrad_group(jobs, tile=tile, math=math)
    for j, (gw, gb), (s1, b1), (B, Cin, Cout, H, W, k, s) in zip(jobs, refs, singles, cases):
        mine = j["slabs"].sum(0).reshape(Cout, k, k, Cin).permute(0, 3, 1, 2)
        assert rel_err(mine, gw) < 2e-5
        assert rel_err(j["dbias"].sum(0), gb) < 2e-5
        assert torch.equal(j["slabs"], s1) and torch.equal(j["dbias"], b1)


H16_CASES = [
    # B, Cin, Cout, H, W, k, stride, tile
    (2, 256, 256, 30, 40, 3, 1, 1),
    (2, 256, 256, 30, 40, 3, 1, 0x203),
    (4, 256, 256, 80, 80, 3, 1, 0x203),      # tail split
    (2, 64, 256, 20, 24, 1, 1, 0),
    (1, 512, 512, 15, 20, 3, 1, 0x3003),     # forced split-K
    (2, 128, 128, 18, 22, 3, 2, 0),          # strided (parity-class dgrad)
    (1, 256, 21, 15, 20, 3, 1, 0),           # predictor head: fp32 output from bf16 inputs
]


@pytest.mark.parametrize("case", H16_CASES)
def test_conv_bf16_storage(K, case):
    """bf16-storage mode of the implicit-GEMM kernel (tile_override 0x800): bf16 activations / weights / residual /
    mask in HBM, v_mfma_f32_32x32x16_bf16, fp32 accumulate, bf16 (or, +0x10000, fp32) output.  Reference = fp64
    convolution of the same bf16 values; a bf16 output may differ from the rounded reference by one bf16 step where
    the fp32 accumulation order moves the value across a rounding boundary."""
    B, Cin, Cout, H, W, k, s, tile = case
    g = torch.Generator().manual_seed(sum(case))
    bf = torch.bfloat16
    x = torch.randn(B, Cin, H, W, generator=g).to(bf)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(bf)
    bias = torch.randn(Cout, generator=g)
    pad = k // 2
    y_ref = F.conv2d(x.double(), w.double(), bias.double(), stride=s, padding=pad)
    Ho, Wo = y_ref.shape[2:]
    res = torch.randn(B, Cout, Ho, Wo, generator=g).to(bf)
    out_ref = F.relu(y_ref + res.double())
    dev = "cuda"
    lv = K.Levels([(H, W)], B)
    geom = K.ConvGeom(lv, Cin, Cout, k, s, pad)
    xr, wf = to_rows(x).to(dev), fold_w(w).to(dev)
    small = Cout < 32
    y = torch.empty(B * Ho * Wo, Cout, device=dev, dtype=torch.float32 if small else bf)
    K.conv_fwd(geom, xr, wf, bias.to(dev), y, addend=to_rows(res).to(dev), relu=True,
               tile=tile | 0x800 | (0x10000 if small else 0))
    got = from_rows(y.float(), B, Ho, Wo).double().cpu()
    if small:
        assert rel_err(got, out_ref) < 1e-5
    else:
        err = (got - out_ref).abs()
        assert (err <= out_ref.abs() * 2.0 ** -7 + 1e-6).all(), float((err / out_ref.abs().clamp_min(1e-3)).max())
        assert rel_err(got, out_ref) < 5e-3
    if small:
        return
    # dgrad with ReLU mask, bf16 in / out
    dy = torch.randn(B, Cout, Ho, Wo, generator=g).to(bf)
    gx = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), stride=s, padding=pad)
    wft = w.permute(1, 2, 3, 0).reshape(Cin, k * k, Cout).contiguous().to(dev)
    mask = to_rows(torch.randn(B, Cin, H, W, generator=g)).to(bf).to(dev)
    dx = torch.empty(B * H * W, Cin, device=dev, dtype=bf)
    K.conv_dgrad(geom, to_rows(dy).to(dev), wft, dx, mask=mask, tile=tile | 0x800)
    gx_m = gx * (from_rows(mask.float().cpu(), B, H, W) > 0)
    got = from_rows(dx.float(), B, H, W).double().cpu()
    err = (got - gx_m).abs()
    assert (err <= gx_m.abs() * 2.0 ** -7 + 1e-6).all()
    assert rel_err(got, gx_m) < 5e-3


@pytest.mark.parametrize("case", [(2, 256, 256, 30, 40, 3, 1, 0), (1, 128, 512, 33, 21, 1, 1, 0), (2, 128, 128, 18, 22, 3, 2, 0),
                                  (1, 256, 21, 15, 20, 3, 1, 0), (2, 512, 256, 16, 20, 1, 1, 1 << 4),
                                  (4, 256, 256, 80, 80, 3, 1, 0)])
def test_wgrad_bf16_storage(K, case):
    """bf16-storage wgrad (flags bit 1): bf16 dy / x, transposing LDS reads, fp32 slabs == fp64 wgrad of the same
    bf16 values (fp32 accumulation tolerance), incl. the fused bias column sums, ragged pixel counts, strided and
    small-Cout cases."""
    import ctypes as C
    from radet_amd import _lib
    B, Cin, Cout, H, W, k, s, wflags = case
    g = torch.Generator().manual_seed(sum(case) + 7)
    bf = torch.bfloat16
    x = torch.randn(B, Cin, H, W, generator=g).to(bf)
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    dy = torch.randn(B, Cout, Ho, Wo, generator=g).to(bf)
    gw = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, k, k), dy.double(), stride=s, padding=pad)
    dev = "cuda"
    lv = K.Levels([(H, W)], B)
    geom = K.ConvGeom(lv, Cin, Cout, k, s, pad)
    ld = (Cout + 7) // 8 * 8
    dyr = torch.zeros(B * Ho * Wo, ld, device=dev, dtype=bf)
    dyr[:, :Cout] = to_rows(dy).to(dev)
    S = geom.nsplit
    slabs = torch.empty(S, Cout, k * k, Cin, device=dev)
    bp = torch.empty(S, Cout, device=dev)
    xr = to_rows(x).to(dev)        # keep alive: a temporary would be freed (and its block reused) before the launch
    _lib.call("radet_conv2d_wgrad", K._ptr(dyr), K._ptr(xr), K._ptr(slabs), K._ptr(bp), K._ptr(geom.fwd_table),
              geom.lout.rows, Cin, Cout, ld, k, k, S, 2 | wflags, K._stream())
    gw_mine = slabs.sum(0).reshape(Cout, k, k, Cin).permute(0, 3, 1, 2)
    assert rel_err(gw_mine, gw) < 2e-5
    assert rel_err(bp.sum(0), dy.double().sum((0, 2, 3))) < 2e-5


def test_conv_multilevel(K):
    """Five pyramid levels in one launch == per-level convs."""
    B, Cch = 2, 256
    hw = [(12, 16), (6, 8), (3, 4), (2, 2), (1, 1)]
    g = torch.Generator().manual_seed(5)
    w = torch.randn(Cch, Cch, 3, 3, generator=g) / (Cch * 9) ** 0.5
    xs = [torch.randn(B, Cch, h, ww, generator=g) for h, ww in hw]
    lv = K.Levels(hw, B)
    geom = K.ConvGeom(lv, Cch, Cch, 3, 1, 1)
    xr = torch.cat([to_rows(x) for x in xs]).cuda()
    y = torch.empty(lv.rows, Cch, device="cuda")
    K.conv_fwd(geom, xr, fold_w(w).cuda(), None, y)
    for i, x in enumerate(xs):
        r0, r1 = lv.level_rows(i)
        ref = F.conv2d(x.double(), w.double(), padding=1)
        assert rel_err(from_rows(y[r0:r1], B, *hw[i]), ref) < 1e-5


def test_gather_tables_survive_cache_eviction(K, monkeypatch):
    """The table cache is bounded; a geometry must keep its own tables (and dgrad parity classes) alive when the cache
    drops them -- a table freed under a launch that is still running on a side stream gets recycled by the allocator
    (found as a GPU memory fault late in a full test run, once > 1024 tables had been built)."""
    monkeypatch.setattr(K._TABLE_CACHE, "cap", 2)
    g0 = K.ConvGeom(K.Levels([(9, 7)], 2), 32, 32, 3, 2, 1)
    t_f, t_b, cls = g0.fwd_table, g0.bwd_table, K._strided_dgrad_classes(g0)
    snap = t_f.clone()
    for i in range(6):                                   # push g0's entries out of the cache, recycle freed blocks
        gi = K.ConvGeom(K.Levels([(9 + i, 8)], 2), 32, 32, 3, 1, 1)
        assert gi.fwd_table.numel() and gi.bwd_table.numel()
    assert len(K._TABLE_CACHE) <= 2 and not any(k[1:] == ("f",) + g0._key for k in K._TABLE_CACHE)
    assert g0.fwd_table is t_f and g0.bwd_table is t_b and K._strided_dgrad_classes(g0) is cls
    assert torch.equal(t_f, snap)
    x = torch.randn(g0.lin.rows, 32, device="cuda")
    w = torch.randn(32 * 9 * 32, device="cuda") * 0.05
    y1, y2 = torch.empty(g0.lout.rows, 32, device="cuda"), torch.empty(g0.lout.rows, 32, device="cuda")
    K.conv_fwd(g0, x, w, None, y1)
    g1 = K.ConvGeom(K.Levels([(9, 7)], 2), 32, 32, 3, 2, 1)        # same geometry, rebuilt tables
    K.conv_fwd(g1, x, w, None, y2)
    assert torch.equal(y1, y2)


@pytest.mark.parametrize("shape", [(256, 256, 3, 30, 40, 4), (1024, 256, 1, 30, 40, 4), (512, 2048, 1, 15, 20, 4),
                                   (96, 80, 3, 13, 7, 2), (256, 256, 3, 5, 4, 4)])
def test_streamk_schedule_matches_plain_launch(K, shape):
    """Stream-K (persistent workgroups share the K stages of the launch evenly, cut tiles reduced in the launch in K
    order): same result as the plain launch up to fp32 re-association, bit-identical from run to run, for every
    workgroups-per-CU setting and tile; residual addend + ReLU epilogue and ragged M / N edges included."""
    cin, cout, k, H, W, B = shape
    lv = K.Levels([(H, W)], B)
    g = K.ConvGeom(lv, cin, cout, k, 1, k // 2)
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(lv.rows, cin, generator=gen).cuda()
    w = (torch.randn(cout * k * k * cin, generator=gen) / (cin * k * k) ** 0.5).cuda()
    add = torch.randn(lv.rows, cout, generator=gen).cuda()
    ref = torch.empty(lv.rows, cout, device="cuda")
    K.conv_fwd(g, x, w, None, ref, addend=add, relu=True, tile=3 | (1 << 12), splitk=False)
    scale = float(ref.abs().max())
    for tile in (3, 4, 2):
        for bk in (0, 0x200):
            if bk and cin % 32:
                continue
            for wgs in (1, 2, 3, 4):
                t = tile | bk | (wgs * K.STREAMK)
                y1, y2 = torch.full_like(ref, float("nan")), torch.full_like(ref, float("nan"))
                K.conv_fwd(g, x, w, None, y1, addend=add, relu=True, tile=t)
                K.conv_fwd(g, x, w, None, y2, addend=add, relu=True, tile=t)
                assert torch.equal(y1, y2), (tile, bk, wgs)
                assert float((y1 - ref).abs().max()) <= 2e-5 * scale, (tile, bk, wgs, float((y1 - ref).abs().max()), scale)
    ws = K.splitk_ws()
    assert int(ws[:16384].view(torch.int32).abs().sum()) == 0          # every ticket handed back


def test_stem_maxpool(K):
    g = torch.Generator().manual_seed(1)
    B, H, W = 2, 70, 90
    img = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    bias = torch.randn(64, generator=g)
    ref = F.relu(F.conv2d(img.double(), w.double(), bias.double(), stride=2, padding=3))
    Ho, Wo = ref.shape[2:]
    y = torch.empty(B * Ho * Wo, 64, device="cuda")
    K.stem(img.cuda(), w.permute(0, 2, 3, 1).contiguous().cuda(), bias.cuda(), y, B, H, W)
    assert rel_err(from_rows(y, B, Ho, Wo), ref) < 1e-5
    pref = F.max_pool2d(ref, 3, 2, 1)
    Hp, Wp = pref.shape[2:]
    p = torch.empty(B * Hp * Wp, 64, device="cuda")
    K.maxpool(y, p, B, Ho, Wo, 64)
    assert rel_err(from_rows(p, B, Hp, Wp), F.max_pool2d(from_rows(y.cpu(), B, Ho, Wo).double(), 3, 2, 1)) == 0.0
    # bf16 storage variant, and the headline image size (full 8 x 32 tiles) / a size below one tile
    yh = torch.empty(B * Ho * Wo, 64, device="cuda", dtype=torch.bfloat16)
    K.stem(img.cuda(), w.permute(0, 2, 3, 1).contiguous().cuda(), bias.cuda(), yh, B, H, W)
    assert rel_err(from_rows(yh.float(), B, Ho, Wo), ref) < 1e-2
    for (b2, h2, w2) in ((1, 480, 640), (3, 9, 13)):
        im = torch.randn(b2, 3, h2, w2, generator=g)
        r2 = F.relu(F.conv2d(im.double(), w.double(), bias.double(), stride=2, padding=3))
        y2 = torch.empty(b2 * r2.shape[2] * r2.shape[3], 64, device="cuda")
        K.stem(im.cuda(), w.permute(0, 2, 3, 1).contiguous().cuda(), bias.cuda(), y2, b2, h2, w2)
        assert rel_err(from_rows(y2, b2, r2.shape[2], r2.shape[3]), r2) < 1e-5, (b2, h2, w2)


def test_groupnorm_fwd_bwd(K):
    B, Cch = 2, 256
    hw = [(9, 13), (5, 7), (3, 3), (2, 1), (1, 1)]
    g = torch.Generator().manual_seed(2)
    lv = K.Levels(hw, B)
    xs = [(torch.randn(B, Cch, h, w, generator=g) * 2 + 0.5).double().requires_grad_(True) for h, w in hw]
    gamma = (torch.rand(Cch, generator=g) + 0.5).double().requires_grad_(True)
    beta = (torch.randn(Cch, generator=g) * 0.3).double().requires_grad_(True)
    ys = [F.relu(F.group_norm(x, 32, gamma, beta, 1e-5)) for x in xs]
    dys = [torch.randn(B, Cch, h, w, generator=g) for h, w in hw]
    grads = torch.autograd.grad(ys, xs + [gamma, beta], [d.double() for d in dys])
    dev = "cuda"
    z = torch.cat([to_rows(x.detach().float()) for x in xs]).to(dev)
    y = torch.empty_like(z)
    stats = torch.empty(len(hw) * B * 64, device=dev)
    ws = torch.empty(K.gn_ws_floats(lv), device=dev)
    gm, bt = gamma.detach().float().to(dev), beta.detach().float().to(dev)
    K.gn_relu_fwd(lv, z, gm, bt, y, stats, ws)
    for i in range(len(hw)):
        r0, r1 = lv.level_rows(i)
        assert rel_err(from_rows(y[r0:r1], B, *hw[i]), ys[i].detach()) < 1e-5
    # the paired launch (cls / reg tower of one layer): bit-identical to two single launches, in fp32 and bf16 storage
    z2, gm2, bt2 = z * 0.7 - 0.2, gm * 1.3, bt - 0.1
    y2, stats2, ws2 = torch.empty_like(z), torch.empty_like(stats), torch.empty_like(ws)
    K.gn_relu_fwd(lv, z2, gm2, bt2, y2, stats2, ws2)
    ya, yb, sa, sb = torch.empty_like(z), torch.empty_like(z), torch.empty_like(stats), torch.empty_like(stats)
    K.gn_relu_fwd_pair(lv, (z, gm, bt, ya, sa, ws), (z2, gm2, bt2, yb, sb, ws2))
    assert torch.equal(ya, y) and torch.equal(yb, y2) and torch.equal(sa, stats) and torch.equal(sb, stats2)
    zh, zh2 = z.bfloat16(), z2.bfloat16()
    yh, yh2, yha, yhb = (torch.empty_like(zh) for _ in range(4))
    K.gn_relu_fwd(lv, zh, gm, bt, yh, sa, ws)
    K.gn_relu_fwd(lv, zh2, gm2, bt2, yh2, sb, ws2)
    K.gn_relu_fwd_pair(lv, (zh, gm, bt, yha, sa, ws), (zh2, gm2, bt2, yhb, sb, ws2))
    assert torch.equal(yha, yh) and torch.equal(yhb, yh2)
    dy = torch.cat([to_rows(d) for d in dys]).to(dev)
    dz = torch.empty_like(z)
    dg, db = torch.empty(Cch, device=dev), torch.empty(Cch, device=dev)
    K.gn_relu_bwd(lv, dy, z, stats, gm, bt, dz, dg, db, ws)
    for i in range(len(hw)):
        r0, r1 = lv.level_rows(i)
        assert rel_err(from_rows(dz[r0:r1], B, *hw[i]), grads[i]) < 2e-5
    assert rel_err(dg, grads[-2]) < 2e-5 and rel_err(db, grads[-1]) < 2e-5


def test_upsample_add(K):
    g = torch.Generator().manual_seed(3)
    B, Cch = 2, 256
    for (ho, wo), (hi, wi) in [((60, 80), (30, 40)), ((13, 13), (7, 7)), ((25, 25), (13, 13))]:
        dst = torch.randn(B, Cch, ho, wo, generator=g)
        src = torch.randn(B, Cch, hi, wi, generator=g).double().requires_grad_(True)
        up = F.interpolate(src, size=(ho, wo), mode="nearest")
        ref = dst.double() + up
        d = to_rows(dst).cuda()
        K.upsample_add(d, to_rows(src.detach().float()).cuda(), B, ho, wo, hi, wi, Cch)
        assert rel_err(from_rows(d, B, ho, wo), ref.detach()) < 1e-6
        dd = torch.randn(B, Cch, ho, wo, generator=g)
        gs, = torch.autograd.grad(up, src, dd.double())
        base = torch.randn(B, Cch, hi, wi, generator=g)
        ds = to_rows(base).cuda()
        K.upsample_add_bwd(ds, to_rows(dd).cuda(), B, ho, wo, hi, wi, Cch)
        assert rel_err(from_rows(ds, B, hi, wi), base.double() + gs) < 1e-5


LEVEL_HW = [(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)]
STRIDES = (8, 16, 32, 64, 128)


def synth_head_outputs(seed, B, cls_mean=-2.0):
    g = torch.Generator().manual_seed(seed)
    cls, reg, iou = [], [], []
    for (h, w) in LEVEL_HW:
        cls.append(torch.randn(B, 21, h, w, generator=g) * 1.5 + cls_mean)
        reg.append(torch.relu(torch.randn(B, 4, h, w, generator=g) * 2.0 + 2.5))
        iou.append(torch.randn(B, 1, h, w, generator=g))
    return cls, reg, iou


def flat(ts):
    return torch.cat([to_rows(t) for t in ts])


def pack_targets(golden, tags, dev):
    a = golden("assigner")
    boxes = [a[t + "_boxes"] for t in tags]
    counts = [b.shape[0] for b in boxes]
    off = np.zeros(len(tags) + 1, np.int32)
    off[1:] = np.cumsum(counts)
    return dict(boxes=torch.from_numpy(np.concatenate(boxes)).to(dev),
                labels=torch.from_numpy(np.concatenate([a[t + "_labels"] for t in tags])).to(dev),
                off=torch.from_numpy(off).to(dev),
                p2g=torch.from_numpy(np.stack([a[t + "_p2g"].astype(np.int64) for t in tags])).to(dev),
                pw=torch.from_numpy(np.stack([a[t + "_w"] for t in tags])).to(dev))


def run_head_loss(K, cls, reg, iou, tg, B, scales=None, grad_scale=None, dumps=False):
    dev = "cuda"
    lv = K.Levels(LEVEL_HW, B)
    ld, nl = K.level_desc(lv, STRIDES)
    R = lv.rows
    fc, fr, fi = flat(cls).to(dev), flat(reg).to(dev), flat(iou).reshape(-1).contiguous().to(dev)
    sc = torch.ones(5, device=dev) if scales is None else scales
    out = dict(losses=torch.zeros(3, device=dev), dcls=torch.zeros(R, 21, device=dev), dreg=torch.zeros(R, 4, device=dev),
               diou=torch.zeros(R, device=dev), dsc=torch.zeros(5, device=dev))
    ws = torch.zeros(K.head_loss_ws_ints(R), dtype=torch.int32, device=dev)
    lab = torch.zeros(R, dtype=torch.long, device=dev) if dumps else None
    tgt = torch.zeros(R, 4, device=dev) if dumps else None
    K.head_loss(fc, fr, fi, sc, tg["boxes"], tg["labels"], tg["off"], tg["p2g"], tg["pw"], ld, nl, B, 21, 0.25, 2.0, 2.0,
                1e-6, grad_scale, out["losses"], out["dcls"], 21, out["dreg"], 4, out["diou"], 1, out["dsc"], ws, lab, tgt)
    out["labels"], out["tgt"], out["ws"] = lab, tgt, ws
    return out


def test_head_loss_vs_reference_golden(K, golden):
    g = golden("head_loss")
    cls, reg, iou = synth_head_outputs(7, 2)
    tg = pack_targets(golden, ("g8", "g3"), "cuda")
    o = run_head_loss(K, cls, reg, iou, tg, 2, dumps=True)
    L = o["losses"].cpu().numpy()
    for i, k in enumerate(("loss_cls", "loss_bbox", "loss_iou")):
        assert abs(L[i] - float(g[k])) <= TOL * max(1.0, abs(float(g[k]))), (k, L[i], float(g[k]))
    assert np.array_equal(o["labels"].cpu().numpy(), g["labels"].astype(np.int64))      # index parity: bit-exact
    pos = torch.from_numpy(g["pos"])
    P = int(o["ws"][0].item())
    assert P == pos.numel() and np.array_equal(o["ws"][16:16 + P].cpu().numpy(), g["pos"])
    assert np.array_equal(o["tgt"].cpu()[pos].numpy(), g["bbox_targets_pos"])             # TBLR targets: bit-exact
    assert float(o["tgt"].abs().sum()) == float(np.abs(g["bbox_targets_pos"]).sum())
    dc, dr, di = o["dcls"].cpu(), o["dreg"].cpu(), o["diou"].cpu()
    assert np.allclose(dc[::7].numpy(), g["g_cls_rows"], rtol=TOL, atol=1e-9)
    assert np.allclose(dc[pos].numpy(), g["g_cls_pos"], rtol=TOL, atol=1e-9)
    # the golden gradient is w.r.t. the post-ReLU tensor; ours is w.r.t. the pre-ReLU output (relu'(0) = 0)
    live = (flat(reg)[pos] > 0).numpy()
    assert np.allclose(dr[pos].numpy()[live], g["g_reg_pos"][live], rtol=TOL, atol=1e-8)
    assert float(np.abs(dr[pos].numpy()[~live]).sum()) == 0.0
    assert np.allclose(di[pos].numpy(), g["g_iou_pos"].reshape(-1), rtol=TOL, atol=1e-9)
    outside = torch.ones(dr.shape[0], dtype=torch.bool)
    outside[pos] = False
    assert float(dr[outside].abs().sum()) == 0.0 and float(di[outside].abs().sum()) == 0.0   # zero outside positives
    assert np.isclose(di.double().abs().sum().item(), float(g["g_iou_abs"]), rtol=TOL)


def test_head_loss_no_gt_and_scales(K, golden):
    g = golden("head_loss")
    cls, reg, iou = synth_head_outputs(8, 2)
    dev = "cuda"
    tg = dict(boxes=torch.zeros(1, 4, device=dev), labels=torch.zeros(1, dtype=torch.long, device=dev),
              off=torch.zeros(3, dtype=torch.int32, device=dev), p2g=torch.full((2, 6400), -1, dtype=torch.long, device=dev),
              pw=torch.ones(2, 6400, device=dev))
    o = run_head_loss(K, cls, reg, iou, tg, 2)
    L = o["losses"].cpu().numpy()
    assert abs(L[0] - float(g["e_loss_cls"])) <= TOL * float(g["e_loss_cls"]) and L[1] == 0.0 and L[2] == 0.0
    assert float(o["dreg"].abs().sum()) == 0.0 and float(o["diou"].abs().sum()) == 0.0
    assert np.isclose(o["dcls"].double().abs().sum().item(), float(g["e_g_cls_abs"]), rtol=TOL)
    # Scale parameters + upstream gradient scaling vs the oracle (autograd through relu(scale * u))
    from oracle import model as om
    a = golden("assigner")
    tags = ("g8", "g3")
    cls, reg, iou = synth_head_outputs(11, 2)
    gsc = torch.Generator().manual_seed(4)
    reg_u = [torch.randn(r.shape, generator=gsc) * 2 + 1.5 for r in reg]
    scales = torch.tensor([0.8, 1.1, 0.9, 1.3, 0.7], requires_grad=True)
    for t in cls + reg_u + iou:
        t.requires_grad_(True)
    regs = [F.relu(u * scales[i]) for i, u in enumerate(reg_u)]
    losses, _ = om.head_loss(cls, regs, iou, [torch.from_numpy(a[t + "_boxes"]) for t in tags],
                             [torch.from_numpy(a[t + "_labels"]) for t in tags],
                             [torch.from_numpy(a[t + "_p2g"].astype(np.int64)) for t in tags],
                             [torch.from_numpy(a[t + "_w"]) for t in tags])
    up = torch.tensor([0.5, 2.0, 3.0])
    (losses["loss_cls"] * up[0] + losses["loss_bbox"] * up[1] + losses["loss_iou"] * up[2]).backward()
    tg = pack_targets(golden, tags, dev)
    o = run_head_loss(K, cls, reg_u, iou, tg, 2, scales=scales.detach().to(dev), grad_scale=up.to(dev))
    assert np.allclose(o["losses"].cpu().numpy(), [losses[k].item() for k in ("loss_cls", "loss_bbox", "loss_iou")], rtol=TOL)
    assert np.allclose(o["dsc"].cpu().numpy(), scales.grad.numpy(), rtol=5e-4, atol=1e-7)
    assert rel_err(o["dreg"], flat([u.grad for u in reg_u])) < 5e-4
    assert rel_err(o["dcls"], flat([c.grad for c in cls])) < 5e-4
    assert rel_err(o["diou"], flat([c.grad for c in iou]).reshape(-1)) < 5e-4


VOTE_CFG = dict(type="vote", iou_threshold=0.65, cluster_score=["cls", "iou"], vote_score=["iou", "cls"],
                iou_enable=False, sima=0.025)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_nms_ops_bit_exact(golden, tag):
    """radet.ops API on the GPU == the reference's C++ ops (golden), bit for bit."""
    from radet_amd import ops
    g = golden("nms")
    bx, cl, ct, lb = (torch.from_numpy(g[f"{tag}_{k}"]) for k in ("boxes", "cls", "ctr", "labels"))
    b, l = ops.vote_nms(bx, cl, lb, VOTE_CFG, score_factor=ct)
    assert np.array_equal(l.numpy(), g[tag + "_vote_l"]) and np.array_equal(b.numpy(), g[tag + "_vote_b"])
    b, l = ops.vote_nms(bx, cl, lb, VOTE_CFG, score_factor=ct, max_num=100)
    assert np.array_equal(b.numpy(), g[tag + "_vote_b"][:100])
    b, l = ops.global_vote_nms(bx, cl, lb, VOTE_CFG, score_factor=ct)
    assert np.array_equal(l.numpy(), g[tag + "_gvote_l"]) and np.array_equal(b.numpy(), g[tag + "_gvote_b"])
    ids, num = ops.cluster_nms(bx.numpy(), (cl * ct).numpy(), lb.numpy(), 0.65)
    assert np.array_equal(ids.numpy(), g[tag + "_cl_ids"]) and np.array_equal(num.numpy(), g[tag + "_cl_num"])
    from oracle import nms as onms
    dets, keep = ops.batched_nms(bx, cl * ct, lb, dict(type="nms", iou_threshold=0.5))
    odets, okeep = onms.batched_nms(bx.numpy(), (cl * ct).numpy(), lb.numpy(), 0.5)
    assert np.array_equal(keep.numpy(), okeep) and np.array_equal(dets.numpy(), odets)


def test_nms_edge_cases():
    from radet_amd import ops
    b, l = ops.vote_nms(torch.zeros(0, 4), torch.zeros(0), torch.zeros(0, dtype=torch.long), VOTE_CFG,
                        score_factor=torch.zeros(0))
    assert tuple(b.shape) == (0, 5) and tuple(l.shape) == (0,)
    b, l = ops.vote_nms(torch.tensor([[1., 2., 30., 40.]]), torch.tensor([0.5]), torch.tensor([3]), VOTE_CFG,
                        score_factor=torch.tensor([0.5]))
    assert np.allclose(b.numpy(), [[1, 2, 30, 40, 0.25]]) and l.tolist() == [3]
    # 8192 boxes (capacity), one label, identical boxes -> a single cluster
    n = 8192
    bx = torch.tensor([[10., 10., 50., 60.]]).repeat(n, 1)
    sc = torch.linspace(0.9, 0.1, n)
    b, l = ops.vote_nms(bx, sc, torch.zeros(n, dtype=torch.long), VOTE_CFG, score_factor=torch.ones(n))
    assert b.shape[0] == 1 and np.allclose(b[0, :4].numpy(), [10, 10, 50, 60], atol=1e-3) and abs(b[0, 4].item() - 0.9) < 1e-6


def test_decode_and_nms_vs_reference_golden(K, golden):
    """radet_decode_candidates + radet_nms on synthetic head outputs == reference get_bboxes."""
    g = golden("get_bboxes")
    cls, reg, iou = synth_head_outputs(9, 2, cls_mean=-4.0)
    dev = "cuda"
    B = 2
    lv = K.Levels(LEVEL_HW, B)
    ld, nl = K.level_desc(lv, STRIDES)
    fc, fr, fi = flat(cls).to(dev), flat(reg).to(dev), flat(iou).reshape(-1).contiguous().to(dev)
    cap = 5000
    boxes, scores, ctr = torch.empty(B, cap, 4, device=dev), torch.empty(B, cap, device=dev), torch.empty(B, cap, device=dev)
    labels, count = torch.empty(B, cap, dtype=torch.long, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
    ws = torch.empty(K.decode_ws_bytes(B, 5, 1000), dtype=torch.uint8, device=dev)
    hw = torch.tensor([[480., 640.]] * B, device=dev)
    sf = torch.ones(B, 4, device=dev)
    K.decode_candidates(fc, fr, fi, torch.ones(5, device=dev), ld, nl, B, 21, 0.05, 1000, hw, sf, boxes, scores, ctr, labels,
                        count, ws)
    n_ref = int((torch.cat([to_rows(c) for c in cls]).sigmoid() > 0.05).sum())
    assert abs(int(count.sum()) - min(n_ref, int(count.sum()))) == 0 and int(count.sum()) > 2000
    for mode, name in ((0, "vote"), (1, "global_vote")):
        cs = (scores * ctr).contiguous()
        ob, osc = torch.zeros(B, 100, 4, device=dev), torch.zeros(B, 100, device=dev)
        ol, oc = torch.zeros(B, 100, dtype=torch.long, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
        a0 = torch.zeros(B, cap, dtype=torch.long, device=dev)
        nws = torch.empty(K.nms_ws_bytes(B, cap), dtype=torch.uint8, device=dev)
        K.nms(boxes, cs, cs, labels, count, B, cap, mode, 0.65, False, 0.025, 100, ob, osc, ol, oc, a0, a0.clone(), nws)
        for i in range(B):
            k = int(oc[i])
            ref_b, ref_l = g[f"{name}_{i}_b"], g[f"{name}_{i}_l"]
            assert k == ref_b.shape[0]
            assert np.array_equal(ol[i, :k].cpu().numpy(), ref_l)
            assert np.allclose(ob[i, :k].cpu().numpy(), ref_b[:, :4], rtol=TOL, atol=1e-3)
            assert np.allclose(osc[i, :k].cpu().numpy(), ref_b[:, 4], rtol=TOL, atol=1e-7)


@pytest.mark.parametrize("cls_mean", [-4.0, -2.0, 0.0])
def test_decode_fast_and_multipass_paths_agree(K, cls_mean):
    """decode_kernel has a fast path (candidates compacted once into an LDS list, <= 16384 per level and image) and a
    multi-pass path over the raw scores (denser outputs): both must produce bit-identical candidate lists; at
    cls_mean = 0 half of all scores pass the threshold, so level 0 takes the multi-pass path by itself."""
    import os
    cls, reg, iou = synth_head_outputs(11, 2, cls_mean=cls_mean)
    dev, B, cap = "cuda", 2, 5000
    lv = K.Levels(LEVEL_HW, B)
    ld, nl = K.level_desc(lv, STRIDES)
    fc, fr, fi = flat(cls).to(dev), flat(reg).to(dev), flat(iou).reshape(-1).contiguous().to(dev)
    hw = torch.tensor([[480., 640.]] * B, device=dev)
    outs = []
    for slow in (False, True):
        if slow:
            os.environ["RADET_DECODE_SLOW"] = "1"
        try:
            boxes, scores = torch.zeros(B, cap, 4, device=dev), torch.zeros(B, cap, device=dev)
            ctr, labels = torch.zeros(B, cap, device=dev), torch.zeros(B, cap, dtype=torch.long, device=dev)
            count = torch.zeros(B, dtype=torch.int32, device=dev)
            ws = torch.empty(K.decode_ws_bytes(B, 5, 1000), dtype=torch.uint8, device=dev)
            K.decode_candidates(fc, fr, fi, torch.ones(5, device=dev), ld, nl, B, 21, 0.05, 1000, hw, None, boxes, scores,
                                ctr, labels, count, ws)
            torch.cuda.synchronize()
        finally:
            os.environ.pop("RADET_DECODE_SLOW", None)
        outs.append([t.cpu() for t in (count, boxes, scores, ctr, labels)])
    n_pass = [[int((to_rows(c)[i * h * w:(i + 1) * h * w].sigmoid() > 0.05).sum()) for c, (h, w) in zip(cls, LEVEL_HW)]
              for i in range(B)]
    for i in range(B):
        k = int(outs[0][0][i])
        assert k == int(outs[1][0][i])
        assert abs(k - sum(min(1000, n) for n in n_pass[i])) <= 2          # torch's sigmoid may differ by an ulp at 0.05
        for a, b in zip(outs[0][1:], outs[1][1:]):
            assert torch.equal(a[i, :k], b[i, :k])
        s = outs[0][2][i, :k]
        assert (s > 0.05).all()


ASSIGN_TAGS = ["g0", "g1", "g8", "g8b", "g20", "g3"]


def test_assigner_bit_exact(golden):
    """GPU assigner, whole batch in one launch, == reference LabelAssignment for the same NumPy seeds."""
    from radet_amd.datasets import LabelAssignment
    g = golden("assigner")
    la = LabelAssignment(anchor_generator_cfg=None, neg_threshold=0.2, positive_num=10, adapt_positive_num=False,
                         balance_sample=True)
    boxes, masks, rngs = [], [], []
    for t in ASSIGN_TAGS:
        G = g[t + "_boxes"].shape[0]
        boxes.append(g[t + "_boxes"])
        masks.append(np.unpackbits(g[t + "_masks"], axis=1).reshape(G, 480, 640) if G else np.zeros((0, 480, 640), np.uint8))
        rngs.append(np.random.RandomState(int(g[t + "_npseed"])))
    p2g, pw = la.assign_batch(boxes, masks, (480, 640, 3), rngs=rngs)
    p2g, pw = p2g.cpu().numpy(), pw.cpu().numpy()
    for i, t in enumerate(ASSIGN_TAGS):
        assert np.array_equal(p2g[i], g[t + "_p2g"].astype(np.int64)), t
        assert np.array_equal(pw[i], g[t + "_w"]), t
        # RNG left exactly where the reference leaves it
        probe = np.random.RandomState(int(g[t + "_npseed"]))
        probe.random_sample(int(g[t + "_used"]))
        assert rngs[i].random_sample() == probe.random_sample()


def test_assigner_constructor_options_bit_exact(golden):
    """LabelAssignment(balance_sample=False / multiply_samplepro_for_weight=True / adapt_positive_num=True /
    random_sample_by_distance=False), alone and together, on visible masks (u8) and on graded float maps (the mask-free
    sampler's entry point): == outputs of the reference for the same NumPy seeds, RNG position included."""
    from oracle import synth
    from radet_amd.datasets import LabelAssignment
    from test_oracle import ASSIGN_OPT_TAGS, assigner_opt_case
    g = golden("assigner_opts")
    for t in ASSIGN_OPT_TAGS:
        boxes, labels, maps, npseed, opts = assigner_opt_case(g, t)
        la = LabelAssignment(anchor_generator_cfg=None, neg_threshold=0.2, positive_num=10, **opts)
        rng = np.random.RandomState(npseed)
        p2g, pw = la.assign_batch([boxes], [maps], (480, 640, 3), rngs=[rng])
        assert np.array_equal(p2g[0].cpu().numpy(), g[t + "_p2g"].astype(np.int64)), t
        assert np.array_equal(pw[0].cpu().numpy(), g[t + "_w"]), t
        probe = np.random.RandomState(npseed)
        probe._bit_generator.random_raw(int(g[t + "_used_words"]))
        assert rng.random_sample() == probe.random_sample(), t
    with pytest.raises(NotImplementedError):
        LabelAssignment(ambiguous_sample="max_dis")


def test_assigner_all_masks_empty():
    """No visible pixel: every candidate has p = 1e-8, sum needs NumPy's pairwise order; compare with the oracle."""
    from oracle import assigner as oa
    from radet_amd.datasets import LabelAssignment
    boxes = np.array([[100, 80, 420, 400], [30, 30, 90, 100]], np.float32)
    masks = np.zeros((2, 480, 640), np.uint8)
    la = LabelAssignment(neg_threshold=0.2, positive_num=10, balance_sample=True)
    p2g, pw = la.assign_batch([boxes], [masks], (480, 640, 3), rngs=[np.random.RandomState(77)])
    rp, rw = oa.assign_points(boxes, np.zeros(2, np.int64), masks, (480, 640, 3), rng=np.random.RandomState(77))
    assert np.array_equal(p2g[0].cpu().numpy(), rp) and np.array_equal(pw[0].cpu().numpy(), rw)


def test_adamw_clip_step(K):
    g = torch.Generator().manual_seed(9)
    n = 100003
    p0 = torch.randn(n, generator=g)
    ps = torch.nn.Parameter(p0.clone().double())
    opt = torch.optim.AdamW([ps], lr=4e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    dev = "cuda"
    p, m, v = p0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    parts, gn = torch.zeros(256, device=dev), torch.zeros(1, device=dev)
    for step in range(1, 4):
        gr = torch.randn(n, generator=g) * (3.0 if step == 2 else 0.05)
        ps.grad = gr.clone().double()
        tn = torch.nn.utils.clip_grad_norm_([ps], 35.0)
        opt.step()
        gd = gr.to(dev)
        K.sqnorm_partials(gd, n, parts)
        K.adamw_step(p, gd, m, v, n, 4e-4, (0.9, 0.999), 1e-8, 0.05, step, 35.0, 1.0, parts, gn)
        assert abs(gn.item() - tn.item()) <= 1e-5 * tn.item()
        assert rel_err(p, ps.detach()) < 1e-6


def test_anchors(golden):
    from radet_amd.core import build_anchor_generator
    ag = build_anchor_generator(dict(type="AnchorGenerator", ratios=[1.0], octave_base_scale=8, scales_per_octave=1,
                                     strides=[8, 16, 32, 64, 128]))
    g = golden("anchors")
    for tag in ("a480x640", "a800x800"):
        sizes = [tuple(int(v) for v in s) for s in g[tag + "_sizes"]]
        a = torch.cat(ag.grid_anchors(sizes, device="cuda")).cpu().numpy()
        assert np.array_equal(a, g[tag])


def test_cabi_error_codes(K):
    """INTEGRATION.md "Error behaviour": bad arguments return -1 (RadetHipError on the Python side), nothing is launched."""
    from radet_amd import _lib
    dev = "cuda"
    lv = K.Levels([(8, 8)], 1)
    g = K.ConvGeom(lv, 24, 32, 1, 1, 0)                                   # Cin % 16 != 0
    x, w, y = torch.zeros(64, 24, device=dev), torch.zeros(32, 1, 24, device=dev), torch.zeros(64, 32, device=dev)
    with pytest.raises(_lib.RadetHipError):
        K.conv_fwd(g, x, w, None, y)
    g2 = K.ConvGeom(lv, 48, 32, 1, 1, 0)                                  # bf16 storage needs Cin % 32 == 0
    with pytest.raises(_lib.RadetHipError):
        K.conv_fwd(g2, torch.zeros(64, 48, device=dev, dtype=torch.bfloat16), torch.zeros(32, 1, 48, device=dev, dtype=torch.bfloat16),
                   None, torch.zeros(64, 32, device=dev, dtype=torch.bfloat16))
    n = 70000                                                             # NMS capacity is 65536 candidates per image
    z = torch.zeros(1, n, device=dev)
    with pytest.raises(_lib.RadetHipError):
        K.nms(torch.zeros(1, n, 4, device=dev), z, z, torch.zeros(1, n, dtype=torch.long, device=dev),
              torch.tensor([n], dtype=torch.int32, device=dev), 1, n, 0, 0.5, False, 0.025, 100, torch.zeros(1, 100, 4, device=dev),
              torch.zeros(1, 100, device=dev), torch.zeros(1, 100, dtype=torch.long, device=dev),
              torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, n, dtype=torch.long, device=dev),
              torch.zeros(1, n, dtype=torch.long, device=dev), torch.zeros(1024, dtype=torch.uint8, device=dev))
    with pytest.raises((_lib.RadetHipError, KeyError)):
        K.mask_transform(torch.zeros(1, 8, 8, dtype=torch.uint8, device=dev), (4, 4), (8, 8))   # pad target smaller than the resized mask
    with pytest.raises(_lib.RadetHipError):                               # wgrad: dy row stride must hold whole float4s
        _lib.call("radet_conv2d_wgrad", K._ptr(y), K._ptr(x), K._ptr(y), None, K._ptr(g2.fwd_table), 64, 48, 32, 30, 1, 1, 1, 0,
                  K._stream())


# ---------------------------------------------------------------------------------------------- plane operands
PLANE_CASES = [
    # B, Cin, Cout, H, W, k, stride, tile
    (2, 64, 64, 20, 24, 1, 1, 3),
    (2, 64, 256, 20, 24, 1, 1, 2),
    (1, 128, 128, 17, 23, 3, 1, 3),            # ragged last M tile, image-border taps
    (2, 256, 256, 30, 40, 3, 1, 1),
    (2, 256, 256, 30, 40, 3, 1, 5),            # 128 x 128, 8 waves
    (2, 256, 256, 30, 40, 3, 1, 6),            # 256 x 128, 8 waves
    (1, 512, 512, 15, 20, 3, 1, 0x3003),       # forced split-K = 3
    (4, 256, 256, 80, 80, 3, 1, 0x20001),      # 3 LDS stages, tail split
    (1, 1024, 256, 15, 20, 1, 1, 0x2003),      # forced split-K = 2
]


@pytest.mark.parametrize("case", PLANE_CASES)
def test_plane_operand_conv(K, case):
    """radet_conv2d_igemm with x / w as bf16 plane triples (tile_override 0x2000000): the plane split is exact, and the
    conv is held to the same fp64 reference and tolerance as the fp32 paths (forward with bias + residual + ReLU, dgrad
    with mask); the 3x3 / 256-channel cases also run the plane-operand all-taps wgrad."""
    B, Cin, Cout, H, W, k, s, tile = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout, generator=g)
    pad = k // 2
    y_ref = F.conv2d(x.double(), w.double(), bias.double(), stride=s, padding=pad)
    Ho, Wo = y_ref.shape[2:]
    res = torch.randn(B, Cout, Ho, Wo, generator=g)
    out_ref = F.relu(y_ref + res.double())
    dy = torch.randn(B, Cout, Ho, Wo, generator=g)
    gx = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), stride=s, padding=pad)

    dev = "cuda"
    lv = K.Levels([(H, W)], B)
    geom = K.ConvGeom(lv, Cin, Cout, k, s, pad)
    xr = to_rows(x).to(dev)
    xp = K.Planes.from_float(xr)
    assert torch.equal(xp.to_float(), xr)                       # hi + mid + lo == x, bit for bit
    wp = K.Planes.from_float(fold_w(w).reshape(Cout * k * k, Cin).to(dev))
    y = torch.empty(B * Ho * Wo, Cout, device=dev)
    K.conv_fwd(geom, xp, wp, bias.to(dev), y, addend=to_rows(res).to(dev), relu=True, tile=tile)
    assert rel_err(from_rows(y, B, Ho, Wo), out_ref) < 1e-5
    dyr = to_rows(dy).to(dev)
    dyp = K.Planes.from_float(dyr)
    wtp = K.Planes.from_float(w.permute(1, 2, 3, 0).reshape(Cin * k * k, Cout).contiguous().to(dev))
    dx = torch.empty(B * H * W, Cin, device=dev)
    mask = to_rows(torch.randn(B, Cin, H, W, generator=g)).to(dev)
    K.conv_dgrad(geom, dyp, wtp, dx, mask=mask, tile=tile)
    gx_m = gx * (from_rows(mask.cpu(), B, H, W) > 0)
    assert rel_err(from_rows(dx, B, H, W), gx_m) < 1e-5
    if k == 3 and Cin % 32 == 0 and Cout == 256:
        gw = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), stride=s, padding=pad)
        S = geom.nsplit
        slabs = torch.empty(S, Cout, k * k, Cin, device=dev)
        K.conv_wgrad(geom, dyp, xp, slabs)
        gw_mine = slabs.sum(0).reshape(Cout, k, k, Cin).permute(0, 3, 1, 2)
        assert rel_err(gw_mine, gw) < 2e-5


def test_plane_outputs_of_groupnorm_and_fold(K):
    """GroupNorm + ReLU forward / backward with plane outputs == the fp32 outputs bit for bit (the split is exact)."""
    dev = "cuda"
    lv = K.Levels([(12, 16), (6, 8), (3, 4)], 2)
    R = lv.rows
    g = torch.Generator().manual_seed(5)
    z = torch.randn(R, 256, generator=g).to(dev)
    gam, bet = (torch.rand(256, generator=g) + 0.5).to(dev), (torch.randn(256, generator=g) * 0.1).to(dev)
    stats = torch.empty(len(lv) * lv.B * 64, device=dev)
    ws = torch.empty(K.gn_ws_floats(lv), device=dev)
    y = torch.empty_like(z)
    K.gn_relu_fwd(lv, z, gam, bet, y, stats, ws)
    yp, y2 = K.Planes(R, 256, device=dev), torch.empty_like(z)
    K.gn_relu_fwd_p(lv, z, gam, bet, y2, yp, stats, ws)
    assert torch.equal(y2, y) and torch.equal(yp.to_float(), y)
    yq, zb = K.Planes(R, 256, device=dev), torch.randn(R, 256, generator=g).to(dev)
    yb = torch.empty_like(z)
    stats_b, ws_b = torch.empty_like(stats), torch.empty_like(ws)
    K.gn_relu_fwd_pair_p(lv, (z, gam, bet, None, yp, stats, ws), (zb, gam, bet, yb, yq, stats_b, ws_b))
    K.gn_relu_fwd(lv, zb, gam, bet, y2, stats_b, ws_b)
    assert torch.equal(yp.to_float(), y) and torch.equal(yq.to_float(), y2) and torch.equal(yb, y2)
    dy = torch.randn(R, 256, generator=g).to(dev)
    dz, dg, db = torch.empty_like(z), torch.empty(256, device=dev), torch.empty(256, device=dev)
    K.gn_relu_bwd(lv, dy, z, stats, gam, bet, dz, dg, db, ws)
    dzp, dg2, db2 = K.Planes(R, 256, device=dev), torch.empty(256, device=dev), torch.empty(256, device=dev)
    K.gn_relu_bwd_p(lv, dy, z, stats, gam, bet, None, dzp, dg2, db2, ws)
    assert torch.equal(dzp.to_float(), dz) and torch.equal(dg2, dg) and torch.equal(db2, db)
    # special values survive the split: zeros, tiny and huge magnitudes, a full 24-bit significand (inputs in fp32's own
    # denormal range are outside the contract: their low planes are bf16 denormals)
    v = torch.tensor([0.0, -0.0, 1e-30, -3.4e38, 1.0 + 2.0 ** -23, 2.0 ** -120, 65504.0, -(2.0 - 2.0 ** -23)] * 4, device=dev).reshape(1, 32)
    back = K.Planes.from_float(v).to_float()
    assert torch.equal(back, v), (back - v)


def test_split_k_in_launch_reduction_under_uneven_load(K):
    """The in-launch split-K reduction (write-through partial tiles, arrival ticket, per-wave agent acquire in the last
    arriver) repeated 3000 times while another stream keeps the chip unevenly busy, with a tile count that puts the
    splits of one tile on different XCDs: every launch must reproduce the first result bit for bit, for the fp32 kernels
    and the plane-operand kernel (8 waves)."""
    dev = "cuda"
    B, Cin, Cout, H, W = 1, 512, 192, 15, 20                 # 5 x 3 = 15 tiles of 64 x 64: 15 % 8 != 0
    lv = K.Levels([(H, W)], B)
    geom = K.ConvGeom(lv, Cin, Cout, 3, 1, 1)
    g = torch.Generator().manual_seed(17)
    x = torch.randn(lv.rows, Cin, generator=g).to(dev)
    w = (torch.randn(Cout * 9, Cin, generator=g) * 0.02).to(dev)
    side = torch.cuda.Stream()
    noise_a = torch.randn(4096, 4096, device=dev)
    for mode in ("fp32", "x3", "planes"):
        geom.x3 = mode == "x3"
        xs, ws = (K.Planes.from_float(x), K.Planes.from_float(w)) if mode == "planes" else (x, w.view(-1))
        tile = (5 if mode == "planes" else 3) | 0x5000       # forced split-K = 5
        y0, y = torch.empty(lv.rows, Cout, device=dev), torch.empty(lv.rows, Cout, device=dev)
        K.conv_fwd(geom, xs, ws, None, y0, tile=tile)
        torch.cuda.synchronize()
        bad = 0
        for it in range(3000):
            if it % 50 == 0:
                with torch.cuda.stream(side):                # bursts of unrelated work: uneven load, dirty L2 lines
                    noise_a.mul_(1.0001)
            K.conv_fwd(geom, xs, ws, None, y, tile=tile)
            if it % 100 == 99:
                bad += int(not torch.equal(y, y0))
        torch.cuda.synchronize()
        assert bad == 0 and torch.equal(y, y0), (mode, bad)


@pytest.mark.parametrize("case", [(2, 256, 256, 30, 40, 3, 1, 2), (1, 128, 512, 33, 21, 1, 1, 2), (2, 128, 128, 18, 22, 3, 2, 1),
                                  (2, 512, 256, 16, 20, 1, 1, 1), (4, 1024, 256, 15, 20, 1, 1, 2), (3, 96, 160, 17, 23, 3, 1, 2)])
def test_wgrad_one_tap_on_plane_pairs(K, case):
    """conv_wgradq_kernel (radet_conv2d_wgrad_s flags 0x1000 | 0x200 | 0x40): the one-tap weight gradient with dy / x as fp16
    plane pairs (transposing LDS reads) == fp64 wgrad, incl. the bias column sums, ragged pixel counts / channel tiles,
    strided and 1 x 1 (no gather table) cases, both tiles."""
    B, Cin, Cout, H, W, k, s, tile = case
    g = torch.Generator().manual_seed(sum(case) + 3)
    x = torch.randn(B, Cin, H, W, generator=g).clamp_min(-0.5)
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    dy = torch.randn(B, Cout, Ho, Wo, generator=g) * 1e-2
    gw = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, k, k), dy.double(), stride=s, padding=pad)
    geom = K.ConvGeom(K.Levels([(H, W)], B), Cin, Cout, k, s, pad)
    geom.wgrad_pair_flags = 0x40 | (tile << 4)
    xq, dyq = K.Planes.from_float(to_rows(x).cuda(), kind="h2"), K.Planes.from_float(to_rows(dy).cuda(), kind="h2")
    for S in (1, 3):
        geom.nsplit = S
        slabs = torch.full((S, Cout, k * k, Cin), float("nan"), device="cuda")
        bp = torch.full((S, Cout), float("nan"), device="cuda")
        K.conv_wgrad(geom, dyq, xq, slabs, bp)
        assert rel_err(slabs.sum(0).reshape(Cout, k, k, Cin).permute(0, 3, 1, 2), gw) < 2e-5, S
        assert rel_err(bp.sum(0), dy.double().sum((0, 2, 3))) < 2e-5, S


@pytest.mark.parametrize("case", [(2, 64, 64, 20, 24, 3, 3), (1, 128, 128, 17, 23, 3, 8), (2, 256, 64, 12, 16, 3, 0x2003), (2, 64, 128, 9, 11, 1, 7)])
def test_pairs_only_tensors_of_a_bottleneck_block(K, case):
    """Round 6 (Engine.po): a tensor that only conv GEMMs and a ReLU mask read exists ONLY as fp16 plane pairs.
    (1) a forward launch with y = NULL writes nothing but the pairs (bound-scaled), raises the true-amax slot and the pairs
    equal those of the launch that also writes fp32; (2) a dgrad launch reads its ReLU mask from a pair tensor: bit-identical
    to the same launch with the fp32 mask; (3) a dgrad launch writes its output only as pairs, scaled by
    amax(dy) * (largest input-channel L1 norm of the folded weights, from radet_fold_weights: RadetConvDesc.w_l1t): the slot
    equals the host's column sums, the bound bounds, the pairs reproduce the fp32 dgrad to 2^-22; (4) a dgrad launch on pair
    operands with a pair mask matches the fp64 chain like the fp32 path."""
    import ctypes as C
    from radet_amd import _lib
    B, Ci, Co, H, W, k, tile = case
    gen = torch.Generator().manual_seed(sum(case) + 3)
    dev = "cuda"
    pad = k // 2
    x = torch.randn(B, Ci, H, W, generator=gen)
    w = torch.randn(Co, Ci, k, k, generator=gen) / (Ci * k * k) ** 0.5
    bias = torch.randn(Co, generator=gen)
    geom = K.ConvGeom(K.Levels([(H, W)], B), Ci, Co, k, 1, pad)
    geom.x3 = "h2"
    R = B * H * W
    xr, wf, br = to_rows(x).to(dev), fold_w(w).to(dev), bias.to(dev)
    l1, ba = K.new_amax(dev), K.new_amax(dev)
    l1[0] = int(torch.tensor(float(wf.abs().reshape(Co, -1).sum(1).max()), dtype=torch.float32).view(torch.int32))
    K.absmax(torch.cat([br, torch.zeros((4 - Co % 4) % 4, device=dev)]), ba)
    # (1) pairs-only forward output
    y = torch.empty(R, Co, device=dev)
    q_both, q_only = K.Planes(R, Co, device=dev, kind="h2"), K.Planes(R, Co, device=dev, kind="h2")
    ys = K.new_amax(dev)
    key = K.register_amax(y, ys)
    K.conv_fwd(geom, xr, wf, br, y, relu=True, tile=tile, yq=q_both, wmeta=(l1, ba))
    q_only.true_amax = K.new_amax(dev)
    q_only.t.fill_(float("nan"))
    K.conv_fwd(geom, xr, wf, br, None, relu=True, tile=tile, yq=q_only, wmeta=(l1, ba))
    assert torch.equal(q_only.t.view(torch.int16), q_both.t.view(torch.int16))
    assert K.amax_value(q_only.true_amax) == K.amax_value(ys) == float(y.abs().max())
    assert K.amax_value(q_only.amax) == K.amax_value(q_both.amax)
    K.unregister_amax([key])
    # (2) ReLU mask from a pair tensor == fp32 mask (y is a ReLU output: zeros and positives)
    dy = torch.randn(R, Co, generator=gen).to(dev)
    act = torch.relu(torch.randn(R, Ci, generator=gen)).to(dev)
    act[::7] = 0.0
    act[3::11] *= 2.0 ** -30                                                   # tiny positives: still positive in the pairs
    actq = K.Planes.from_float(act, kind="h2")
    dx_a, dx_b = torch.empty(R, Ci, device=dev), torch.empty(R, Ci, device=dev)
    wtf = wf.reshape(Co, k * k, Ci).permute(2, 1, 0).contiguous().reshape(-1)     # [Ci][tap][Co]: the dgrad's weight operand
    K.conv_dgrad(geom, dy, wtf, dx_a, mask=act, tile=tile)
    K.conv_dgrad(geom, dy, wtf, dx_b, mask=actq, tile=tile)
    assert torch.equal(dx_a, dx_b)
    assert bool(((dx_a != 0) <= (act > 0)).all())
    # (3) dgrad output only as pairs, bound from the fold's input-channel L1 slot
    d = _lib.RadetConvDesc()
    wp = w.to(dev).contiguous()
    wf2, wft2, bf2 = torch.empty(Co * k * k * Ci, device=dev), torch.empty(Ci * k * k * Co, device=dev), torch.empty(Co, device=dev)
    wa, l1t = K.new_amax(dev), K.new_amax(dev)
    d.w, d.wf, d.wft, d.bias_f = wp.data_ptr(), wf2.data_ptr(), wft2.data_ptr(), bf2.data_ptr()
    d.cout, d.cin, d.kh, d.kw, d.nsplit, d.eps = Co, Ci, k, k, 1, 1e-5
    d.w_amax, d.w_l1t = wa.data_ptr(), l1t.data_ptr()
    table = torch.frombuffer(bytearray(bytes(d)), dtype=torch.uint8).to(dev)
    K.fold_weights(table, 1)
    want = float(w.abs().sum(dim=(0, 2, 3)).max())
    assert abs(K.amax_value(l1t) - want) <= 1e-5 * want, (K.amax_value(l1t), want)
    assert torch.equal(wft2, wtf)
    kw = K.register_amax(wft2, wa)
    dq = K.Planes(R, Ci, device=dev, kind="h2")
    dq.true_amax = K.new_amax(dev)
    K.conv_dgrad(geom, dy, wft2, None, mask=act, tile=tile, yq=dq, wmeta=(l1t, None))
    K.conv_dgrad(geom, dy, wft2, dx_b, mask=act, tile=tile)
    bound, amax = K.amax_value(dq.amax), float(dx_b.abs().max())
    assert K.amax_value(dq.true_amax) == amax and amax <= bound <= 256 * amax, (amax, bound)
    back = dq.to_float()
    tol = dx_b.abs() * 2.0 ** -22 + bound * 2.0 ** -36
    assert bool(((back - dx_b).abs() <= tol).all()), float(((back - dx_b).abs() / tol).max())
    K.unregister_amax([kw])
    # (4) dgrad on pair operands + pair mask against fp64
    dyq = K.Planes.from_float(dy, kind="h2")
    wtq = K.Planes.from_float(wft2.reshape(Ci * k * k, Co), kind="h2")
    ref = torch.nn.grad.conv2d_input((B, Ci, H, W), w.double(), from_rows(dy.cpu(), B, H, W).double(), stride=1, padding=pad)
    ref = ref * (from_rows(act.cpu(), B, H, W) > 0)
    for t in [(tile & 0xF000) | 3] + ([(tile & 0xF000) | 7] if Co % 64 == 0 else []):     # 4-wave pair tile; K-divided pair tile (round 6)
        dx_a.fill_(float("nan"))
        K.conv_dgrad(geom, dyq, wtq, dx_a, mask=actq, tile=t)
        assert rel_err(from_rows(dx_a, B, H, W), ref) < 1e-5, hex(t)
    # (5) the K-divided pair tile forward == the 4-wave pair tile's result to fp32 summation order, incl. split-K and the epilogue
    if Ci % 64 == 0:
        xq, wq = K.Planes.from_float(xr, kind="h2"), K.Planes.from_float(wf.reshape(Co * k * k, Ci), kind="h2")
        y3, y7 = torch.empty(R, Co, device=dev), torch.empty(R, Co, device=dev)
        y_ref = torch.relu(F.conv2d(x.double(), w.double(), bias.double(), padding=pad))
        K.conv_fwd(geom, xq, wq, br, y3, relu=True, tile=3)
        for t in (7, 0x2007):
            y7.fill_(float("nan"))
            K.conv_fwd(geom, xq, wq, br, y7, relu=True, tile=t)
            assert rel_err(from_rows(y7, B, H, W), y_ref) < 1e-5, hex(t)
            assert float((y7 - y3).abs().max()) <= 2e-6 * float(y3.abs().max())


@pytest.mark.parametrize("case", [(2, 64, 64, 20, 24, 1, 1, 3), (2, 256, 128, 18, 22, 3, 2, 2), (1, 128, 256, 17, 23, 3, 1, 1),
                                  (2, 512, 128, 15, 20, 1, 1, 0x3003), (1, 64, 256, 9, 11, 1, 1, 7), (2, 128, 128, 12, 16, 3, 1, 8)])
def test_conv_epilogue_writes_plane_pair_copy(K, case):
    """RadetScales.yq: a conv launch (any fp32-tensor tile, here the in-register fp16 hi / lo ones, and plane-pair inputs)
    writes its output once more as fp16 plane pairs, scaled by a bound it can form before it starts,
    amax(x) * L1max(w) + max|bias| + amax(addend).  The bound bounds (and is not wildly loose), the pairs reproduce y to
    2^-22 of every element, and a second conv that reads them (no operand split in its K loop) matches the fp64 chain like
    the fp32 path does -- incl. split-K, K-divided and strided producers and a residual + ReLU epilogue."""
    B, Cin, Cout, H, W, k, s, tile = case
    g = torch.Generator().manual_seed(sum(case) + 11)
    dev = "cuda"
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout, generator=g)
    pad = k // 2
    y_ref = F.conv2d(x.double(), w.double(), bias.double(), stride=s, padding=pad)
    Ho, Wo = y_ref.shape[2:]
    res = torch.randn(B, Cout, Ho, Wo, generator=g)
    out_ref = F.relu(y_ref + res.double())
    geom = K.ConvGeom(K.Levels([(H, W)], B), Cin, Cout, k, s, pad)
    geom.x3 = "h2"
    xr, wf = to_rows(x).to(dev), fold_w(w).to(dev)
    resr, br = to_rows(res).to(dev), bias.to(dev)
    l1 = K.new_amax(dev)
    l1[0] = int(torch.tensor(float(wf.abs().reshape(Cout, -1).sum(1).max()), dtype=torch.float32).view(torch.int32))
    ba = K.new_amax(dev)
    K.absmax(torch.cat([br, torch.zeros((4 - Cout % 4) % 4, device=dev)]), ba)
    y = torch.empty(B * Ho * Wo, Cout, device=dev)
    yq = K.Planes(B * Ho * Wo, Cout, device=dev, kind="h2")
    ys = K.new_amax(dev)
    key = K.register_amax(y, ys)
    for xin, win in ((xr, wf), (K.Planes.from_float(xr, kind="h2"), K.Planes.from_float(wf.reshape(Cout * k * k, Cin), kind="h2"))):
        if K._isp(xin):
            xin.true_amax = K.amax_slot(xr, compute=True)
            t = (tile & 0xF000) | (3 if (tile & 0xFF) >= 7 else tile & 0xFF)          # (no K-divided tile for plane operands)
        else:
            t = tile
        y.fill_(float("nan")); yq.t.zero_(); ys.zero_()
        K.conv_fwd(geom, xin, win, br, y, addend=resr, relu=True, tile=t, yq=yq, wmeta=(l1, ba))
        assert rel_err(from_rows(y, B, Ho, Wo), out_ref) < 1e-5
        bound, amax = K.amax_value(yq.amax), float(y.abs().max())
        assert K.amax_value(ys) == amax and amax <= bound <= 64 * amax + 8.0, (amax, bound)
        back = yq.to_float()
        tol = y.abs() * 2.0 ** -22 + bound * 2.0 ** -36
        assert bool(((back - y).abs() <= tol).all()), float(((back - y).abs() / tol).max())
    K.unregister_amax([key])
    # the consumer: a 1 x 1 conv on the pairs against the fp64 chain
    C2 = 96
    w2 = torch.randn(C2, Cout, 1, 1, generator=g) / Cout ** 0.5
    ref2 = F.conv2d(out_ref, w2.double())
    g2 = K.ConvGeom(K.Levels([(Ho, Wo)], B), Cout, C2, 1, 1, 0)
    y2 = torch.empty(B * Ho * Wo, C2, device=dev)
    K.conv_fwd(g2, yq, K.Planes.from_float(fold_w(w2).reshape(C2, Cout).to(dev), kind="h2"), None, y2, tile=3)
    assert rel_err(from_rows(y2, B, Ho, Wo), ref2) < 1e-5


@pytest.mark.parametrize("hw,B,cin,cout", [([(12, 16), (6, 8), (3, 4)], 2, 64, 160),      # ragged last output tile, small levels
                                           ([(15, 20), (8, 10), (4, 5)], 3, 32, 128),      # widths that divide no 16-pixel stage
                                           ([(9, 7), (1, 1)], 2, 32, 256),                 # a row shorter than the halo, a 1 x 1 level
                                           ([(60, 80)], 1, 256, 256)])                     # a tower level of the headline config
def test_all_taps_wgrad_variants_are_bit_identical(K, monkeypatch, hw, B, cin, cout):
    """The all-taps weight gradient on fp16 plane pairs of a unit-stride 3 x 3 conv, three kernels (radet_conv2d_wgrad_s flags):
    conv_wgrad9q_kernel (two stage buffers), conv_wgrad9d_kernel (+0x2000, round 6, the default: five buffers, loads four stages
    ahead, the gather table of the pixel split in LDS as 16-bit differences, bias sums by inline-asm LDS reads) and
    conv_wgrad9r_kernel (+0x4000, an experiment: the nine taps read shifted windows of three row segments, pixels whose tap is
    padding zeroed in registers).  They form the same products in the same order, so slabs and bias column sums are equal bit
    for bit -- on multi-level geometries whose rows are shorter than / do not divide the 16-pixel stages, with ragged pixel
    splits and splits shorter than the pipeline -- and agree with the fp64 weight gradient (reference: the autograd of
    `F.conv2d` in `dense_heads/atss_head.py:118-145`)."""
    dev = "cuda"
    lv = K.Levels(hw, B)
    geom = K.ConvGeom(lv, cin, cout, 3, 1, 1)
    g = torch.Generator().manual_seed(lv.rows + cin)
    x = torch.randn(lv.rows, cin, generator=g).to(dev)
    dy = (torch.randn(lv.rows, cout, generator=g) * 1e-3).to(dev)
    xp, dyp = K.Planes.from_float(x, kind="h2"), K.Planes.from_float(dy, kind="h2")
    gw = torch.zeros(cout, cin, 3, 3, dtype=torch.float64)
    off = 0
    for (h, w) in hw:                                                 # fp64 reference, level by level
        n = B * h * w
        xi = x[off:off + n].cpu().double().reshape(B, h, w, cin).permute(0, 3, 1, 2)
        di = dy[off:off + n].cpu().double().reshape(B, h, w, cout).permute(0, 3, 1, 2)
        gw += torch.nn.grad.conv2d_weight(xi, gw.shape, di, padding=1)
        off += n
    for S in (1, 3, 7, 64):
        geom.nsplit = min(S, -(-lv.rows // 16))
        S = geom.nsplit
        out = {}
        for name, deep, windows in (("two buffers", False, False), ("deep", True, False), ("windows", False, True)):
            monkeypatch.setattr(K, "WGRAD9_DEEP", deep)
            monkeypatch.setattr(K, "WGRAD9_WINDOWS", windows)
            slabs = torch.full((S, cout, 9, cin), float("nan"), device=dev)
            bp = torch.full((S, cout), float("nan"), device=dev)
            K.conv_wgrad(geom, dyp, xp, slabs, bp)
            out[name] = (slabs, bp)
        for name in ("deep", "windows"):
            assert torch.equal(out[name][0], out["two buffers"][0]) and torch.equal(out[name][1], out["two buffers"][1]), (name, S)
        gw_mine = out["deep"][0].sum(0).reshape(cout, 3, 3, cin).permute(0, 3, 1, 2)
        assert rel_err(gw_mine, gw) < 2e-5, S
