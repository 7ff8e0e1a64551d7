"""Static execution plan of the detector hot path on one MI355X.

The network structure is fixed (ResNet bottlenecks -> FPN -> shared-weight head), so instead of a
tracing compiler or per-op autograd nodes the engine holds an explicit forward program and a
hand-written reverse program over pre-allocated NHWC buffers; every step is a fixed sequence of
HIP launches on the current stream (hipGraph-capturable: no allocation, no host sync).

Layout decisions (MI355X-first):
  * activations: fp32 NHWC rows [B*H*W, C]; the five pyramid levels live in ONE row-concatenated
    buffer so that each shared-weight head conv / GroupNorm is a single launch over B*6400 rows;
  * parameters: torch Parameters in state-dict layout (OIHW) are folded once per step into OHWI
    (+ BN scale) and a transposed copy for dgrad by one batched kernel; wgrad writes split-K slabs
    that a second batched kernel reduces, un-folds (BN gamma/beta grads) and re-lays-out to OIHW.

Reference call sites this replaces: ResNet.forward (radet/models/backbones/resnet.py:622-637),
FPN.forward (necks/fpn.py:170-221), ATSSHead.forward / RADetHead.forward_single
(dense_heads/atss_head.py:100-145, radet_head.py:27-30) and their autograd backward.
"""
import contextlib
import ctypes as C
import os

import torch

from . import _lib
from . import kernels as K
from .kernels import ConvGeom, Levels

ARCH = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}


def conv_out_hw(h, w, k, s, p):
    return (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1


class _StemSplits:
    """What the slab / descriptor bookkeeping needs from a conv geometry, for the stem (which has its own kernels)"""

    def __init__(self, nsplit):
        self.nsplit = nsplit


class Conv:
    """One convolution layer: parameters by name, folded buffers, geometry-bound slabs."""

    def __init__(self, name, cin, cout, k, stride, pad, bn=None, bias=False, trainable=True, dgrad=True):
        self.name, self.cin, self.cout, self.k, self.stride, self.pad = name, cin, cout, k, stride, pad
        self.bn, self.bias, self.trainable, self.need_dgrad = bn, bias, trainable, dgrad
        self.wft_ld, self.wft_off, self.wft_shared = 0, 0, None
        self.grouped = False          # weight gradient issued through the stage's grouped launch (Engine._plan_wgrad_groups)
        self.geom = None
        self.wf = self.wft = self.bias_f = self.slabs = self.dbias_partials = None

    @property
    def wsize(self):
        return self.cout * self.cin * self.k * self.k


class Engine:
    def __init__(self, params, grads, depth=50, num_classes=21, frozen_stages=1, strides=(8, 16, 32, 64, 128),
                 stacked_convs=4, feat=256, math=None, watch=None):
        """params / grads: dict name -> device tensor (reference state-dict names; grads only for
        trainable parameters, same shapes).  math (env RADET_MATH):
          "fp32" (default)  fp32 tensors and fp32-accurate arithmetic.  The conv GEMMs form their products on the 16-bit
                            matrix cores: every fp32 operand is scaled by an exact power of two (from the tensor's
                            tracked largest magnitude) and split into two fp16 numbers hi + 2^-11 lo, and
                            hi hi' + 2^-11 (hi lo' + lo hi') is accumulated in fp32 by three v_mfma_f32_32x32x16_f16 per
                            K = 16 step (error against fp64 at or below the native fp32 MFMA's, 16x its MAC rate per
                            plane product).  RADET_X3=bf16 selects the round-2 scheme (exact three-way bf16 split, 6 of
                            the 9 plane products); RADET_X3=0 or "fp32-mfma" selects v_mfma_f32_32x32x2_f32;
          "fp32-mfma"       the same with the native fp32 matrix instruction;
          "bf16"            conv operands rounded to bf16 on their way into the matrix cores, fp32 accumulate, fp32
                            tensors / GroupNorm / loss / optimizer;
          "bf16-storage"    bf16 activations / folded weights / activation gradients in HBM (the mixed precision of
                            BASELINE config 3)."""
        math = math or os.environ.get("RADET_MATH", "fp32")
        assert math in ("fp32", "fp32-mfma", "bf16", "bf16-storage"), math
        self.watch = watch or {}                  # name -> tensors whose version counters guard the folded weights
        self._watched = {}
        self.math = 1 if math == "bf16" else 0
        # "bf16-storage": activations, folded weights and activation gradients are bf16 tensors in HBM
        # (v_mfma_f32_32x32x16_bf16, fp32 accumulate); head outputs, loss, statistics, weight gradients, master
        # weights and optimizer stay fp32
        self.h16 = math == "bf16-storage"
        x3mode = os.environ.get("RADET_X3", "h2")
        self.x3 = math == "fp32" and x3mode != "0"
        self.h2 = self.x3 and x3mode not in ("bf16", "b3")      # fp16 hi / lo arithmetic (3 plane products instead of 6)
        # Pair copies in the backbone (round 5): the forward launches of the bottleneck blocks and the lateral convs read their
        # x operand as fp16 plane pairs that the PRODUCER's epilogue wrote next to the fp32 tensor (scaled by a bound it can
        # form before it starts, conv_common.h ConvPtrs::yq), and their weights as pairs from the fold -- no operand split in
        # those K loops (25-30 % less time per launch alone, tools/bench_h2.py).  RADET_PAIRS=0: split in registers.
        # The pair copy is a second 4-byte write per output element, and in the step that costs more than the K loops save:
        # measured (r50 640x480 bs 4, same box, ms per step) 8.92 without, 9.10 from layer2 on (RADET_PAIRS_FROM), 8.99 from
        # layer3, 8.96 from layer4, 9.14 with layer1 -- OFF by default (RADET_PAIRS=1 selects it; the kernels, the bound logic
        # and the tests stay).  [A first measurement showed 8.73: the first pair stage was reading a copy nobody had written
        # -- zeros -- and the matrix cores clock higher on zeros.  Timings are only comparable on live data.]
        self.pairs = self.h2 and os.environ.get("RADET_PAIRS", "0") == "1"
        # Pairs-ONLY tensors in the bottleneck blocks (round 6; built, measured, OFF by default -- RADET_PAIRS_ONLY=1 selects it):
        # a tensor that nothing but conv GEMMs and a ReLU mask read is written by its producer's epilogue as fp16 plane pairs
        # and NOT as fp32 (as the towers' activations have been since round 3): the same 4 bytes per element leave the CU, the
        # consumers' K loops run without an operand split, and unlike RADET_PAIRS=1 there is no second write.  In a stride-1
        # bottleneck block: o1 = relu(bn1(conv1 x)) is read by conv2 (forward and weight gradient) and as the ReLU mask of
        # conv2's dgrad; d_o2 = dL/d(o2) is read by conv2's dgrad and weight gradient; conv2's folded weights exist as pairs
        # in both orientations.  The residual path (block inputs / outputs, d_pre) stays fp32, and so do o2 / d_o1 (their
        # weight gradients pair them with an fp32 tensor).  Strided blocks keep fp32 tensors (parity-class dgrad launches).
        # What it measured (r50 640 x 480 bs 4, per launch, serialised, tools/prof_layers.py, pairs-only against fp32 tensors):
        #   conv2 forward   64 ch 34.8 / 48.4 us, 128 ch 39.5 / 47.9, 256 ch 47.7 / 42.6, 512 ch 48.6 / 50.5
        #   conv2 dgrad     128 ch 40.0 / 48.0, 256 ch 48.4 / 41.7, 512 ch 50.1 / 49.2
        #   conv2 wgrad     (one-tap pair kernel) 128 ch 65.7 / 46.8, 256 ch 50.9 / 47.6, 512 ch 57.3 / 54.5
        #   producers       +2-4 us per launch for the pair epilogue (conv1 forward, conv3 dgrad)
        # i.e. the K-divided in-register tiles of round 5 beat the 4-wave pair tiles from K = 2304 on; a K-divided PAIR tile
        # (tile 7 on plane operands) then gives 40.6 / 43.3 us forward and 41.0 / 44.0 us dgrad at 256 / 512 ch -- 2-14 % under
        # the in-register tile: these launches are latency-bound, not split-bound.  The step does not move: 8.33-8.43 ms with
        # every stride-1 block (or those of <= 128 planes: RADET_PAIRS_ONLY_MAX) on pairs against 8.30-8.45 without, same
        # box.  The kernels, the bound logic and the tests stay.
        self.po = self.h2 and not self.pairs and os.environ.get("RADET_PAIRS_ONLY", "0") == "1"
        self.po_max_planes = int(os.environ.get("RADET_PAIRS_ONLY_MAX", "512"))
        self.pairs_from = int(os.environ.get("RADET_PAIRS_FROM", "2"))          # first ResNet stage (1-based) that reads pairs
        if self.x3 and "RADET_TOWER_MODE" not in os.environ:
            self.tower_mode = "pairbwd"
        # plane operands for the head towers (RADET_P3=0: split in the GEMMs' registers as everywhere else): the tensors only
        # conv GEMMs read -- tower activations, their gradients, the towers' folded weights -- are stored as bf16 plane triples
        # by their producers (GroupNorm kernels, fold), and the tower GEMMs run without any operand split in their K loops
        self.p3 = self.x3 and self.tower_mode == "pairbwd" and feat % 32 == 0 and os.environ.get("RADET_P3", "1") != "0"
        self.math_name = math
        self.act_dtype = torch.bfloat16 if self.h16 else torch.float32
        self.p, self.g = params, grads
        self.depth, self.num_classes, self.frozen_stages = depth, num_classes, frozen_stages
        # Stream budget.  This device runs FOUR HIP streams of one process well; with a fifth one IN USE during the step it falls
        # off a cliff: 9.98 -> 13.8 ms with main + side + second weight-gradient stream + tower chain + the process group's
        # internal stream (`tools/bench_dp1.py`; GPU_MAX_HW_QUEUES does not move it; three weight-gradient streams without
        # any collective: 13.2 ms).  Hence: the three extra streams are shared by all engines of a process, decode / NMS and
        # graph capture borrow the tower-chain stream (idle at inference), and the gradient exchange runs as synchronous
        # collectives ON the tower-chain stream (idle from the head's backward pass to the next step), runtime.GradReducer.
        self.strides, self.stacked_convs, self.feat = tuple(strides), stacked_convs, feat
        self.dev = next(iter(params.values())).device
        self.convs = []
        self._build_layers()
        self._alloc_folded()
        self.geo_key = None
        self.table = None

    # ------------------------------------------------------------------ profiling hook
    # own kernel symbol for the head-tower GEMM family (see conv_igemm.hip) + its measured-best tile:
    # 64x64 block tile with a 32-deep K step (bench_conv.py: 98.7 vs 88.8 TFLOP/s for the heuristic pick)
    TOWER_TAG = 0x100 | 0x200 | 3        # backward: 64 x 64 tile, K step 32
    TOWER_TAG_FWD = 0x100 | 0x200 | 2    # forward: 128 x 64 tile (2 accumulators per wave), K step 32

    # plane-operand tower tiles with three LDS stages (144 KiB; experiment switch)
    tower_stages = K.STAGES3 if os.environ.get("RADET_TOWER_STAGES3", "0") == "1" else 0
    tower_tile = int(os.environ.get("RADET_TOWER_TILE", "6"))       # 6: 256 x 128 (one workgroup per CU), 5: 128 x 128 (two)
    if os.environ.get("RADET_TOWER_ROWPAIRS", "0") == "1":          # (K.ROWPAIRS: 128-byte pieces, both planes of a row per load)
        tower_tile |= 0x80000

    def _ttile(self, c, bwd=False, tag=True, pair=True):
        """tile_override of a tower conv launch + profiling tag.  Forward: a fixed, measured tile with a 32-deep K step
        (the forward launches are grouped cls + reg pairs, which the single-conv timing of the autotuner does not
        represent: it picked K step 16 for bf16 storage, 138 instead of 102 us per launch): 128 x 64, two accumulators
        per wave -- with the accumulators in VGPRs it fits 3 waves per SIMD: 120.5 vs 116.5 TFLOP/s for 64 x 64 in fp32,
        +0.6 % / +1.5 % of the step in the bf16-storage / bf16-math modes; the fp32 launches run alone on the device and
        take 3 LDS stages.  Backward: 64 x 64 in fp32 (128 x 64 / 128 x 128 measured equal next to the wgrad streams),
        autotuned in the bf16 modes."""
        fp32 = not self.math and not self.h16
        if self.p3:
            # plane operands: 256 x 128 tiles, 8 waves (one workgroup per CU owns its LDS: 2 x 72 KiB of stages) for the
            # grouped cls + reg launches, 128 x 128 / 8 waves for a single tower GEMM (tools/bench_p3.py)
            return (6 if pair else 5) | (0x100 if tag else 0) | self.tower_stages
        if fp32 and self.x3:
            # products from bf16 planes: the operand split is VALU work per fragment, so the tile with the most MFMAs per
            # fragment wins -- 128 x 128 (4 accumulators per wave), 2 LDS stages, forward (178 vs 158 TFLOP/s fp32-equivalent
            # for 128 x 64) and backward (-0.15 ms per step)
            t = 0x200 | 1
        elif bwd:
            t = (self.TOWER_TAG & ~0x100) if fp32 else (self.tower_bwd_tile_h16 or c.geom.bwd_tile)
        else:
            t = (self.TOWER_TAG_FWD & ~0x100) | (K.STAGES3 if fp32 else 0)
            if not fp32 and self.tower_fwd_tile_h16:
                t = self.tower_fwd_tile_h16
        return t | (0x100 if tag else 0)
    tower_fwd_tile_h16 = int(os.environ.get("RADET_TOWER_FWD_TILE", "0"), 0)     # bf16 modes: tower forward tile (0: 128 x 64, K step 32)
    # bf16 modes: tower dgrad tile.  The tuner times a launch ALONE and picks 64 x 64 / K step 16 for M = 25 600, K = 2304; in the
    # step the two towers' dgrads run next to each other and to the weight gradients, where 128 x 128 / K step 32 is the faster
    # one: bf16-storage step 5.35 -> 5.11 ms (round 6, same box: 0x203 5.24, 0x202 5.18, 0x1 5.16).  0: the tuner's pick
    tower_bwd_tile_h16 = int(os.environ.get("RADET_TOWER_BWD_TILE", "0x201"), 0)
    tower_events = None  # when a list: (start, end) torch.cuda.Event pairs around every tower GEMM launch
    _pfx_ready = None    # (image key, buffer set, event) of a frozen prefix computed ahead of its step (prefetch_prefix)

    def _tower_launch(self, fn, *args, **kw):
        ev = self.tower_events
        if ev is None:
            return fn(*args, **kw)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn(*args, **kw)
        e.record()
        ev.append((s, e))

    def tower_gemm_flops(self):
        """Algorithmic FLOPs of ONE tower GEMM launch: 2 * (B * points) * 256 * (9 * 256)."""
        return 2.0 * self.R * self.feat * 9 * self.feat

    # ------------------------------------------------------------------ structure
    def _add(self, conv):
        self.convs.append(conv)
        return conv

    def _build_layers(self):
        fs = self.frozen_stages
        if fs < 0 and self.h16:
            # the reference trains conv1 / bn1 when frozen_stages = -1 (resnet.py:572-588): built for the fp32-tensor modes
            # (max-pool backward + stem weight gradient read fp32 activations); silently leaving the stem's gradients at zero
            # would let AdamW's weight decay shrink it
            raise NotImplementedError("ResNet(frozen_stages=-1) (trainable stem) with bf16 storage is not implemented on MI355X: "
                                      "every RADet config freezes the stem (frozen_stages >= 0; the BOP configs use 1)")
        self.stem = self._add(Conv("backbone.conv1", 3, 64, 7, 2, 3, bn="backbone.bn1", trainable=fs < 0, dgrad=False))
        self.stages = []
        inpl = 64
        for li, nb in enumerate(ARCH[self.depth]):
            planes = 64 * 2 ** li
            train = (li + 1) > fs
            blocks = []
            for b in range(nb):
                pfx = f"backbone.layer{li + 1}.{b}"
                stride = 2 if (b == 0 and li > 0) else 1
                # gradient w.r.t. the block input is needed unless that input comes from a frozen stage / the stem
                in_dgrad = train and not (b == 0 and li <= fs)
                blk = dict(
                    c1=self._add(Conv(pfx + ".conv1", inpl, planes, 1, 1, 0, bn=pfx + ".bn1", trainable=train, dgrad=in_dgrad)),
                    c2=self._add(Conv(pfx + ".conv2", planes, planes, 3, stride, 1, bn=pfx + ".bn2", trainable=train, dgrad=train)),
                    c3=self._add(Conv(pfx + ".conv3", planes, planes * 4, 1, 1, 0, bn=pfx + ".bn3", trainable=train, dgrad=train)),
                    ds=None, stride=stride, train=train,
                    po=self.po and stride == 1 and planes % 32 == 0 and planes <= self.po_max_planes)   # o1 / d_o2 only as plane pairs
                if b == 0:
                    blk["ds"] = self._add(Conv(pfx + ".downsample.0", inpl, planes * 4, 1, stride, 0, bn=pfx + ".downsample.1",
                                               trainable=train, dgrad=in_dgrad))
                inpl = planes * 4
                blocks.append(blk)
            self.stages.append(blocks)
        f = self.feat
        self.lat = [self._add(Conv(f"neck.lateral_convs.{i}.conv", c, f, 1, 1, 0, bias=True)) for i, c in enumerate((512, 1024, 2048))]
        self.fpn = [self._add(Conv(f"neck.fpn_convs.{i}.conv", f, f, 3, 1 if i < 3 else 2, 1, bias=True)) for i in range(5)]
        self.cls_tower = [self._add(Conv(f"bbox_head.cls_convs.{i}.conv", f, f, 3, 1, 1)) for i in range(self.stacked_convs)]
        self.reg_tower = [self._add(Conv(f"bbox_head.reg_convs.{i}.conv", f, f, 3, 1, 1)) for i in range(self.stacked_convs)]
        self.pred_cls = self._add(Conv("bbox_head.atss_cls", f, self.num_classes, 3, 1, 1, bias=True))
        self.pred_reg = self._add(Conv("bbox_head.atss_reg", f, 4, 3, 1, 1, bias=True))
        self.pred_iou = self._add(Conv("bbox_head.atss_centerness", f, 1, 3, 1, 1, bias=True))
        # dgrad of the small predictors runs on zero-padded K (GEMM K must be a multiple of 16)
        kq = 32 if self.h16 else 16           # bf16 storage: K in 32-channel steps, 16-byte aligned dy columns
        self.cls_pad = ((self.num_classes + kq - 1) // kq) * kq
        self.ri_pad = kq                      # reg (cols 0-3) + iou share one padded gradient buffer
        self.iou_col = 8 if self.h16 else 4
        self.pred_cls.wft_ld, self.pred_cls.wft_off = self.cls_pad, 0
        self.pred_reg.wft_ld, self.pred_reg.wft_off = self.ri_pad, 0
        self.pred_iou.wft_ld, self.pred_iou.wft_off = self.ri_pad, self.iou_col
        self.pred_iou.wft_shared = self.pred_reg

    def _alloc_folded(self):
        dev = self.dev
        n_wf = sum(c.wsize for c in self.convs)
        n_b = sum(c.cout for c in self.convs)
        self.wf_arena = torch.zeros(n_wf, device=dev, dtype=self.act_dtype)
        self.stem_wf = torch.zeros(self.convs[0].wsize, device=dev)      # the (VALU) stem kernel always reads fp32 weights
        self.bias_arena = torch.zeros(n_b, device=dev)
        n_wft = 0
        for c in self.convs:
            if c.need_dgrad and c.wft_shared is None:
                ld = c.wft_ld or c.cout
                n_wft += c.cin * c.k * c.k * ld
        self.wft_arena = torch.zeros(n_wft, device=dev, dtype=self.act_dtype)
        # amax slots of the folded weights (fp16 hi / lo arithmetic): one per conv, written by radet_fold_weights
        self.w_amax = K.new_amax(dev, len(self.convs))
        self.w_l1 = K.new_amax(dev, len(self.convs)) if (self.pairs or self.po) else None   # largest channel L1 norm / largest |bias_f|
        self.b_amax = K.new_amax(dev, len(self.convs)) if (self.pairs or self.po) else None
        self.w_l1t = K.new_amax(dev, len(self.convs)) if self.po else None          # largest INPUT-channel L1 norm (dgrad bounds)
        self._w_amax_keys = []
        o_w = o_b = o_t = 0
        towers = (self.cls_tower + self.reg_tower) if self.p3 else []
        po_c1 = [blk["c1"] for st in self.stages for blk in st if blk["po"]]
        po_c2 = [blk["c2"] for st in self.stages for blk in st if blk["po"]]
        po_c3 = [blk["c3"] for st in self.stages for blk in st if blk["po"] and blk["train"]]
        towers = towers + po_c2                   # conv2 of a pairs-only block: folded weights as pairs in both orientations
        pkind = "h2" if self.h2 else "b3"
        for ci, c in enumerate(self.convs):
            c.w_amax = self.w_amax[ci]
            c.wfq = None
            c.wmeta = (self.w_l1[ci], self.b_amax[ci] if (c.bn or c.bias) else None) if (self.pairs or c in po_c1) else None
            c.wmeta_t = (self.w_l1t[ci], None) if c in po_c3 else None          # (dgrad launches that write pairs)
            c.w16 = (3 if self.h2 else 2) if c in towers else (1 if (self.h16 and c is not self.convs[0]) else 0)
            c.wf = self.wf_arena[o_w:o_w + c.wsize] if c is not self.convs[0] else self.stem_wf
            o_w += c.wsize
            c.bias_f = self.bias_arena[o_b:o_b + c.cout]
            o_b += c.cout
            if c.need_dgrad:
                if c.wft_shared is not None:
                    c.wft = c.wft_shared.wft
                else:
                    n = c.cin * c.k * c.k * (c.wft_ld or c.cout)
                    c.wft = self.wft_arena[o_t:o_t + n]
                    o_t += n
            if c.w16 >= 2:            # planes: rows (o, tap) x Cin and (c, tap) x Cout (include/radet_hip.h, "planes" / "plane pairs")
                c.wf = K.Planes(c.cout * c.k * c.k, c.cin, device=dev, kind=pkind, amax=c.w_amax)
                c.wft = K.Planes(c.cin * c.k * c.k, c.cout, device=dev, kind=pkind, amax=c.w_amax) if c.need_dgrad else None
            elif self.h2 and c is not self.convs[0]:
                if self.pairs and c.name.startswith(("backbone.layer", "neck.lateral")) and c.cin % 32 == 0 and \
                        (not c.name.startswith("backbone.layer") or int(c.name[14]) >= self.pairs_from):
                    c.wfq = K.Planes(c.cout * c.k * c.k, c.cin, device=dev, kind="h2", amax=c.w_amax)
                self._w_amax_keys.append(K.register_amax(c.wf, c.w_amax))
                if c.need_dgrad and c.wft_shared is None:
                    self._w_amax_keys.append(K.register_amax(c.wft, c.w_amax))

    # ------------------------------------------------------------------ geometry-dependent plan
    # Geometry plans: everything that depends on (B, H, W) -- activation / gradient buffers, conv geometries and their
    # tuned tiles, slab arenas, the descriptor table -- is kept per geometry in a small LRU, so a harness that alternates
    # e.g. training at B = 4 with validation at B = 1 neither reallocates nor re-tunes after the first pass.
    max_plans = int(os.environ.get("RADET_MAX_PLANS", "4"))
    _PLAN_ATTRS = ("B", "H", "W", "stem_hw", "pool_hw", "buf", "plv", "R", "gn_ws", "gn_ws2", "ldesc", "nlvl", "loss_ws",
                   "losses", "dscales", "slab_arena", "bp_arena", "table", "_table_keepalive", "max_cout", "_pending_wgrad",
                   "amax_act", "amax_aux", "_amax_keys", "amax_pfx", "_pfx_shapes", "_pfx_sets", "_pfx_active", "_pfx_ready",
                   "plan_id")

    def _snapshot(self):
        return dict(attrs={k: getattr(self, k) for k in self._PLAN_ATTRS},
                    convs=[(c.geom, c.slabs, c.dbias_partials, c.grouped) for c in self.convs],
                    blocks=[[(blk.get("lin"), blk.get("lout")) for blk in blocks] for blocks in self.stages])

    def _restore(self, plan):
        for k, v in plan["attrs"].items():
            setattr(self, k, v)
        for c, (g, sl, bp, gr) in zip(self.convs, plan["convs"]):
            c.geom, c.slabs, c.dbias_partials, c.grouped = g, sl, bp, gr
        for blocks, saved in zip(self.stages, plan["blocks"]):
            for blk, (lin, lout) in zip(blocks, saved):
                blk["lin"], blk["lout"] = lin, lout

    def prepare(self, B, H, W):
        key = (B, H, W)
        if self.geo_key == key:
            return
        if not hasattr(self, "_plans"):
            import collections
            self._plans = collections.OrderedDict()
        if self.geo_key is not None:
            self._plans[self.geo_key] = self._snapshot()          # most recently used goes last
            requested = self._plans.pop(key, None)                # never evict the plan that is being asked for
            while len(self._plans) >= max(self.max_plans, 1):
                # evict the least recently used plan (frees its buffers): nothing on any stream may still be using them
                # when the caching allocator hands the blocks to the new plan
                torch.cuda.synchronize()
                _, old = self._plans.popitem(last=False)
                K.unregister_amax(old["attrs"].get("_amax_keys") or [])
            if requested is not None:
                self._plans[key] = requested
        self.geo_key = key
        if key in self._plans:
            self._restore(self._plans.pop(key))
            return
        self.plans_built = getattr(self, "plans_built", 0) + 1
        self.plan_id = self.plans_built           # identifies this plan's buffers (a launch tape recorded on them: runtime.py)
        dev = self.dev
        self.B, self.H, self.W = B, H, W
        h1, w1 = conv_out_hw(H, W, 7, 2, 3)
        h2, w2 = conv_out_hw(h1, w1, 3, 2, 1)
        self.stem_hw, self.pool_hw = (h1, w1), (h2, w2)
        self.buf = {}
        # amax slots of the activation / gradient buffers (fp16 hi / lo arithmetic): raised by the kernels that write a
        # buffer, zeroed at the start of every forward pass; amax_aux: slots their producers reset themselves (GroupNorm)
        self.amax_act = K.new_amax(dev, 768)
        self.amax_aux = K.new_amax(dev, 32)
        self._amax_keys = []
        n_slot = [0, 0]

        def slot(aux=False):
            i = n_slot[aux]
            n_slot[aux] += 1
            return (self.amax_aux if aux else self.amax_act)[i]

        # Frozen prefix (stem, max-pool, frozen stages): its outputs do not depend on the parameters the step updates, so the
        # prefix of the NEXT batch may run during this step's backward pass (prefetch_prefix).  Its buffers exist twice then
        # (the backward pass still reads this step's set), and their amax slots live in an arena per set that is zeroed by
        # the prefix pass itself, not at the head of the forward pass.
        self.amax_pfx = [K.new_amax(dev, 128), None]
        self._pfx_shapes = {}
        self._pfx_sets = [{}, None]
        self._pfx_active, self._pfx_ready = 0, None
        n_pfx = [0]

        def new(name, rows, ch, dtype=None, prefix=False, twin=False, only=False):
            """twin: a pair copy of the buffer (Planes "h2", written by the producer's epilogue) under name + '@q';
            only: the buffer exists ONLY as plane pairs (self.po) -- buf[name] is the Planes, .true_amax the slot its producer raises"""
            if only:
                q = K.Planes(rows, ch, device=dev, kind="h2")
                if prefix:
                    assert n_pfx[0] < self.amax_pfx[0].shape[0], "frozen prefix: more buffers than amax slots"
                    q.true_amax = self.amax_pfx[0][n_pfx[0]]
                    n_pfx[0] += 1
                    self._pfx_shapes[name] = (rows, ch, "pairs-only", False)
                    self._pfx_sets[0][name] = q
                else:
                    q.true_amax = slot()
                self.buf[name] = q
                return q
            t = torch.empty(rows, ch, device=dev, dtype=dtype or self.act_dtype)
            self.buf[name] = t
            if prefix:
                self._pfx_shapes[name] = (rows, ch, t.dtype, twin and self.pairs)
                self._pfx_sets[0][name] = t
            sl_ = None
            if self.h2 and t.dtype == torch.float32:
                if prefix:
                    assert n_pfx[0] < self.amax_pfx[0].shape[0], "frozen prefix: more buffers than amax slots"
                    sl_ = self.amax_pfx[0][n_pfx[0]]
                    n_pfx[0] += 1
                else:
                    sl_ = slot()
                self._amax_keys.append(K.register_amax(t, sl_, by_storage=True))
            if twin and self.pairs and ch % 32 == 0:
                q = K.Planes(rows, ch, device=dev, kind="h2")
                q.true_amax = sl_                 # (the slot the producer RAISES; q.amax holds the bound its pairs were scaled with)
                self.buf[name + "@q"] = q
                if prefix:
                    self._pfx_sets[0][name + "@q"] = q
            return t

        frozen_prefix = not self.stem.trainable
        new("stem", B * h1 * w1, 64, prefix=frozen_prefix)
        new("pool", B * h2 * w2, 64, prefix=frozen_prefix, twin=self.pairs_from <= 1)
        if self.stem.trainable:           # frozen_stages = -1: gradients w.r.t. the pooled map and the stem's pre-activation
            new("d_pool", B * h2 * w2, 64)
            new("d_stem", B * h1 * w1, 64)
            self.stem.geom = _StemSplits(K.stem_wgrad_splits(B, H, W))
        lv = Levels([(h2, w2)], B)
        for li, blocks in enumerate(self.stages):
            for b, blk in enumerate(blocks):
                pfx = f"l{li + 1}.{b}"
                blk["lin"] = lv
                fz = frozen_prefix and not blk["train"] and all(not bb["train"] for st in self.stages[:li] for bb in st)
                blk["c1"].geom = ConvGeom(lv, blk["c1"].cin, blk["c1"].cout, 1, 1, 0)
                blk["c2"].geom = ConvGeom(lv, blk["c2"].cin, blk["c2"].cout, 3, blk["stride"], 1)
                lo = blk["c2"].geom.lout
                blk["c3"].geom = ConvGeom(lo, blk["c3"].cin, blk["c3"].cout, 1, 1, 0)
                if blk["ds"] is not None:
                    blk["ds"].geom = ConvGeom(lv, blk["ds"].cin, blk["ds"].cout, 1, blk["stride"], 0)
                    new(pfx + ".idt", lo.rows, blk["ds"].cout, prefix=fz)
                blk["lout"] = lo
                tw = li + 1 >= self.pairs_from                  # pair copies where a consumer reads them: inside the stages that
                new(pfx + ".o1", lv.rows, blk["c1"].cout, prefix=fz, twin=tw, only=blk["po"])   # run on pairs, and the block output
                new(pfx + ".o2", lo.rows, blk["c2"].cout, prefix=fz, twin=tw)            # that feeds the first of them
                new(pfx + ".out", lo.rows, blk["c3"].cout, prefix=fz, twin=tw or (li + 2 == self.pairs_from and b == len(blocks) - 1))
                if blk["train"]:
                    new(pfx + ".d_o1", lv.rows, blk["c1"].cout)
                    new(pfx + ".d_o2", lo.rows, blk["c2"].cout, only=blk["po"])
                    new(pfx + ".d_pre", lo.rows, blk["c3"].cout)
                lv = lo
        # FPN on C3..C5
        c_lv = [self.stages[i][-1]["lout"] for i in (1, 2, 3)]
        f = self.feat
        for i in range(3):
            self.lat[i].geom = ConvGeom(c_lv[i], self.lat[i].cin, f, 1, 1, 0)
            new(f"lat{i}", c_lv[i].rows, f)
            new(f"d_lat{i}", c_lv[i].rows, f)
            new(f"d_c{i}", c_lv[i].rows, self.lat[i].cin)
        hw = [c_lv[i].hw[0] for i in range(3)]
        hw.append(conv_out_hw(*hw[-1], 3, 2, 1))
        hw.append(conv_out_hw(*hw[-1], 3, 2, 1))
        self.plv = Levels(hw, B)
        if self.x3:
            K._pred_tiles(self.plv)               # tile table of the predictor convs (built outside any graph capture)
        for i in range(3):
            self.fpn[i].geom = ConvGeom(c_lv[i], f, f, 3, 1, 1)
        self.fpn[3].geom = ConvGeom(self.plv.sub(2), f, f, 3, 2, 1)
        self.fpn[4].geom = ConvGeom(self.plv.sub(3), f, f, 3, 2, 1)
        R = self.plv.rows
        self.R = R
        new("P", R, f)
        new("dP", R, f)
        # gradients w.r.t. P5 / P6 with the stride-2 convs' contributions added: one buffer (and one amax slot) EACH -- the
        # weight-gradient GEMM that reads the first one runs asynchronously, and a slot shared with the second buffer could be
        # raised under it (the operand scale, hence the low bits of the result, would depend on the streams' timing)
        for lvl in (2, 3):
            r0, r1 = self.plv.level_rows(lvl)
            new(f"dP_tmp{lvl}", r1 - r0, f)
        pkind = "h2" if self.h2 else "b3"
        if self.p3:
            self.buf["Pp"] = K.Planes(R, f, device=dev, kind=pkind)      # P as planes: input of both towers' first conv / wgrad
        for t in ("cls", "reg"):
            for i in range(self.stacked_convs):
                new(f"{t}.z{i}", R, f)
                if self.p3 and i < self.stacked_convs - 1:      # read by tower GEMMs only: stored as planes
                    self.buf[f"{t}.y{i}"] = K.Planes(R, f, device=dev, kind=pkind)
                else:
                    new(f"{t}.y{i}", R, f)
                if self.h2:
                    self.buf[f"{t}.zhat{i}"] = slot(aux=True)     # largest normalised magnitude (bounds the backward's dz)
                self.buf[f"{t}.stats{i}"] = torch.empty(len(hw) * B * 64, device=dev)
            new(f"{t}.dy", R, f)
            for nm in (f"{t}.dz0", f"{t}.dz1"):  # two alternating GN-backward outputs: the async wgrad of layer i may still
                if self.p3:                      # read dz[i & 1] while layer i-1 writes the other one
                    self.buf[nm] = K.Planes(R, f, device=dev, kind=pkind)
                else:
                    new(nm, R, f)
            self.buf[f"{t}.dz"] = self.buf[f"{t}.dz0"]
        for c in self.cls_tower + self.reg_tower:
            c.geom = ConvGeom(self.plv, f, f, 3, 1, 1)
            if self.p3 and self.h2:
                # conv_wgrad9q_kernel: 128 output x 32 input channels x 9 taps per workgroup, one workgroup per CU
                c.geom.nsplit = max(1, min(64, 256 // (-(-f // 128) * (f // 32)), (R + 127) // 128))
        for c in (self.pred_cls, self.pred_reg, self.pred_iou):
            c.geom = ConvGeom(self.plv, f, c.cout, 3, 1, 1)
        new("cls", R, self.num_classes, torch.float32)       # head outputs / their gradients: fp32 (loss is fp32)
        new("reg_u", R, 4, torch.float32)
        new("iou", R, 1, torch.float32)
        self.buf["dcls"] = torch.zeros(R, self.cls_pad, device=dev)
        self.buf["dregiou"] = torch.zeros(R, 16, device=dev)
        if self.h2:
            for nm in ("dcls", "dregiou"):
                self._amax_keys.append(K.register_amax(self.buf[nm], slot(), by_storage=True))
        if self.h16:                                         # bf16 copies consumed by the predictors' dgrad / wgrad
            self.buf["dcls16"] = torch.zeros(R, self.cls_pad, device=dev, dtype=torch.bfloat16)
            self.buf["dregiou16"] = torch.zeros(R, self.ri_pad, device=dev, dtype=torch.bfloat16)
        self.gn_ws = torch.empty(K.gn_ws_floats(self.plv), device=dev)
        self.gn_ws2 = torch.empty(K.gn_ws_floats(self.plv), device=dev)
        self.ldesc, self.nlvl = K.level_desc(self.plv, self.strides)
        self.loss_ws = torch.zeros(K.head_loss_ws_ints(R), dtype=torch.int32, device=dev)
        self.losses = torch.zeros(3, device=dev)
        self.dscales = torch.zeros(len(hw), device=dev)
        for c in self.convs:
            if c.geom is not None:
                c.geom.math = self.math
                c.geom.h16 = self.h16
                c.geom.x3 = self.x3
                c.geom.h2 = self.h2
                c.geom.pairs = c.wfq is not None or (self.po and c.w16 == 3 and c.name.startswith("backbone."))
        tune = os.environ.get("RADET_AUTOTUNE", "1") != "0"
        self._plan_wgrad_groups()
        if tune:
            K.load_tune_cache()
            for c in self.convs:          # before the slabs are sized: this chooses the number of pixel splits
                # the head towers keep the launcher's choice (the all-taps kernel): timed alone, the one-tap kernel ties with
                # it on this shape and a noisy pick costs 60 % on the 8 tower launches once they share the chip
                if c.trainable and c.geom is not None and c.cout > 64 and c.cin > 64 and not c.grouped \
                        and c not in self.cls_tower and c not in self.reg_tower:
                    K.autotune_wgrad(c.geom)
        for c in self.convs:                  # backbone / neck weight gradients on plane pairs: the one-tap pair kernel with the
            if c.geom is not None and hasattr(c.geom, "wgrad_pair_flags") and c not in self.cls_tower + self.reg_tower:
                c.geom.wgrad_pair_flags = 0x40 | (c.geom.wgrad_flags & 0x30)        # tuned tile class (128 x 128, else 64 x 64)
        # wgrad slabs / bias partials + descriptor table
        n_slab = sum(c.geom.nsplit * c.wsize for c in self.convs if c.trainable)
        n_bp = sum(c.geom.nsplit * c.cout for c in self.convs if c.trainable)
        self.slab_arena = torch.empty(n_slab, device=dev)
        self.bp_arena = torch.zeros(n_bp, device=dev)
        o_s = o_b = 0
        for c in self.convs:
            if c.trainable:
                n = c.geom.nsplit * c.wsize
                c.slabs = self.slab_arena[o_s:o_s + n]
                o_s += n
                n = c.geom.nsplit * c.cout
                c.dbias_partials = self.bp_arena[o_b:o_b + n]
                o_b += n
        self._build_table()
        if tune:
            towers = self.cls_tower + self.reg_tower
            for c in self.convs:
                # the fp32 tower GEMM has a fixed, measured-best tile (TOWER_TAG); in bf16 math it is tuned like the rest
                if c is not self.stem and c.geom is not None and (self.math or c not in towers):
                    K.autotune(c.geom, need_dgrad=c.need_dgrad)
            K.save_tune_cache()

    # ------------------------------------------------------------------ grouped weight-gradient launches (optional schedule)
    # The weight-gradient GEMMs of the backbone and the neck are independent of each other and off the critical path;
    # each one alone is a short grid (150-1000 workgroups incl. pixel splits).  RADET_WGRAD_GROUP=1 collects them while
    # the dgrad chain of a stage runs and issues ONE grouped launch per stage (radet_conv2d_wgrad_group) on the side
    # stream: the chip stays full with a few long workgroups per conv, so far fewer pixel splits (= slabs) are needed.
    # Measured on r50 640x480 bs 4: wgrad kernel time 5.0 -> 3.3 ms per step (42.8 -> 60 TFLOP/s in-step), slab
    # reduction 0.78 -> 0.50 ms, but the step itself 16.0 -> 16.3 ms: the long-lived workgroups of a group hold their
    # CU slots for ~0.6 ms and the dependent dgrad chain on the main stream waits for slots more often than next to the
    # 50 short launches (HIP stream priorities did not change that).  Hence off by default; kept as a tested option
    # for configurations where the side streams carry more work than the chain (e.g. communication-heavy runs).
    wgrad_group = os.environ.get("RADET_WGRAD_GROUP", "0") == "1"
    wgrad_group_tile = int(os.environ.get("RADET_WGRAD_GROUP_TILE", "128"))
    wgrad_group_slots = int(os.environ.get("RADET_WGRAD_GROUP_SLOTS", "512"))
    wgrad_flush_blocks = int(os.environ.get("RADET_WGRAD_FLUSH_BLOCKS", "0"))   # blocks per grouped launch; 0: the whole stage

    def _plan_wgrad_groups(self):
        for c in self.convs:
            c.grouped = False
        self._pending_wgrad = []
        if not self.wgrad_group or self.h16:
            return
        t = self.wgrad_group_tile
        fb = self.wgrad_flush_blocks
        groups = []
        for blocks in self.stages:                     # launch groups = runs of `fb` blocks in backward order (0: whole stage)
            rev = list(reversed(blocks))
            step = fb if fb > 0 else len(rev)
            for i in range(0, len(rev), step):
                groups.append([c for blk in rev[i:i + step] for c in (blk["c1"], blk["c2"], blk["c3"], blk["ds"])
                               if c is not None and c.trainable])
        groups.append(list(self.lat) + list(self.fpn))
        for grp in groups:
            grp = [c for c in grp if c.cout % t == 0 and c.cin % t == 0]
            if not grp:
                continue
            for c, s in zip(grp, K.group_splits([c.geom for c in grp], tile=t, slots=self.wgrad_group_slots)):
                c.grouped = True
                c.geom.nsplit = s

    def flush_wgrads(self):
        """Issue the collected weight-gradient GEMMs as grouped launches on the side stream (their inputs are complete
        on the current stream at this point)."""
        jobs, self._pending_wgrad = self._pending_wgrad, []
        if not jobs:
            return
        if not self.use_streams:
            K.conv_wgrad_group(jobs, tile=self.wgrad_group_tile, math=self.math)
            return
        side = self._side()
        self._fork(side)
        with torch.cuda.stream(side):
            K.conv_wgrad_group(jobs, tile=self.wgrad_group_tile, math=self.math)

    def _build_table(self):
        n = len(self.convs)
        arr = (_lib.RadetConvDesc * n)()

        def ptr(t):
            return None if t is None else C.c_void_p(t.data_ptr())      # (tensors and K.Planes alike)

        for d, c in zip(arr, self.convs):
            p, g = self.p, self.g
            d.w = ptr(p[c.name + ".weight"])
            d.bias = ptr(p.get(c.name + ".bias")) if c.bias else None
            if c.bn:
                d.bn_gamma, d.bn_beta = ptr(p[c.bn + ".weight"]), ptr(p[c.bn + ".bias"])
                d.bn_mean, d.bn_var = ptr(p[c.bn + ".running_mean"]), ptr(p[c.bn + ".running_var"])
            d.wf, d.wft, d.bias_f = ptr(c.wf), ptr(c.wft) if c.need_dgrad else None, ptr(c.bias_f)
            d.cout, d.cin, d.kh, d.kw = c.cout, c.cin, c.k, c.k
            d.eps = 1e-5
            d.wft_ld, d.wft_off = c.wft_ld, c.wft_off
            d.w16 = c.w16
            d.w_amax = ptr(c.w_amax) if self.h2 else None
            if self.pairs and c is not self.convs[0]:
                d.wfq = ptr(c.wfq)
                d.w_l1, d.bias_amax = ptr(c.wmeta[0]), ptr(self.b_amax[self.convs.index(c)])
            if self.po and c.wmeta is not None:
                d.w_l1, d.bias_amax = ptr(c.wmeta[0]), ptr(self.b_amax[self.convs.index(c)])
            if c.wmeta_t is not None:
                d.w_l1t = ptr(c.wmeta_t[0])
            d.nsplit = c.geom.nsplit if c.geom is not None else 1
            if c.trainable and c.geom is not None:
                d.dwf_slabs, d.dbias_partials = ptr(c.slabs), ptr(c.dbias_partials)
                d.dw = ptr(g[c.name + ".weight"])
                if c.bias:
                    d.dbias = ptr(g[c.name + ".bias"])
                if c.bn:
                    d.dgamma, d.dbeta = ptr(g[c.bn + ".weight"]), ptr(g[c.bn + ".bias"])
        raw = bytes(arr)
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.dev)
        self._table_keepalive = arr
        self.max_cout = max(c.cout for c in self.convs)

    # ------------------------------------------------------------------ per-step parameter transforms
    _fold_event = None

    def fold(self):
        """W' = gamma*rstd*W (+ transposed copies) for every conv.  The leading frozen convs (stem, frozen stages) are
        folded on the current stream; the trainable rest runs on the side stream, concurrently with the frozen part
        of the forward pass, and `_await_fold()` joins it before the first trainable conv."""
        n = len(self.convs)
        nf = 0
        while nf < n and not self.convs[nf].trainable:
            nf += 1
        # Folded weights are kept while their sources are unchanged.  The frozen convs (stem + frozen stages, BN in eval
        # mode) never change between steps: ~0.1 ms per step that sat in front of the stem on the main stream.  The
        # trainable convs change with every optimizer step (`params_changed()`), but not between inference calls.
        # A write through torch (load_state_dict, checkpoint load, replica sync, a torch optimizer) moves the tensors'
        # version counters and is seen here, as is a rebound `.data` (storage pointers are part of the key).  A write
        # THROUGH `.data` (`p.data.copy_()`, an EMA hook) has a version counter of its own and is not: the module-level
        # entry points that can follow user code invalidate explicitly (`RADet.train() / eval()`, `load_state_dict`,
        # `RADet.invalidate_folded_weights()`), raw kernels call `invalidate_fold()`.
        vf, vt = self._part_version(0, nf), self._part_version(nf, n) + (self._param_epoch,)
        do_f = nf > 0 and vf != self._folded[0]
        do_t = nf < n and vt != self._folded[1]
        if do_f:
            self._pfx_ready = None            # a prefix prefetched with the previous frozen weights is stale -- in every plan
            for plan in getattr(self, "_plans", {}).values():
                plan["attrs"]["_pfx_ready"] = None
        tail = self.table[nf * C.sizeof(_lib.RadetConvDesc):]
        if do_t and self.use_streams and nf > 0 and os.environ.get("RADET_FOLD_SIDE", "1") != "0":
            side = self._side()
            self._fork(side)
            with torch.cuda.stream(side):
                K.fold_weights(tail, n - nf)
                ev = self._event()
                K.ev_record(ev)
            self._fold_event = ev
            if do_f:
                K.fold_weights(self.table, nf)
        elif do_f and do_t:
            K.fold_weights(self.table, n)
        elif do_f:
            K.fold_weights(self.table, nf)
        elif do_t:
            K.fold_weights(tail, n - nf)
        self._folded = (vf, vt)
        self._last_fold = (do_f, do_t)          # (a launch tape may only be recorded from a step that folded the trainable part alone)

    def _frozen_conv_count(self):
        nf = 0
        while nf < len(self.convs) and not self.convs[nf].trainable:
            nf += 1
        return nf

    _last_fold = (None, None)
    _folded = (None, None)
    _param_epoch = 0
    _watched = None

    def _part_version(self, lo, hi):
        if os.environ.get("RADET_FOLD_EVERY_CALL"):
            return (object(),)                    # never equal: fold on every call
        key = (lo, hi)
        watched = self._watched.get(key)
        if watched is None:                       # the tensors (arena views + module Parameters) behind convs[lo:hi]
            watched = []
            for c in self.convs[lo:hi]:
                names = [c.name + ".weight"] + ([c.name + ".bias"] if c.bias else [])
                if c.bn:
                    names += [c.bn + sfx for sfx in (".weight", ".bias", ".running_mean", ".running_var")]
                for k in names:
                    if k in self.p:
                        watched += list(self.watch.get(k, (self.p[k],)))
            self._watched[key] = watched
        # (views of one arena share a version counter: a write to any of them re-folds the whole part -- conservative)
        return (tuple(t._version for t in watched), tuple(t.data_ptr() for t in watched), self.h16, self.math)

    def params_changed(self):
        """The trainable parameters were written by a kernel torch does not see (the fused clip + AdamW step)."""
        self._param_epoch += 1

    def invalidate_fold(self):
        """Fold everything again on the next call (after writing parameters through `.data` or a raw kernel)."""
        self._folded = (None, None)

    def _await_fold(self):
        if self._fold_event is not None:
            K.ev_wait(self._fold_event)
            self._fold_event = None

    def unfold(self):
        K.unfold_grads(self.table, len(self.convs), self.max_cout)

    # ------------------------------------------------------------------ forward
    def _block_forward(self, blk, pfx, x, b, xq=None):
        """one bottleneck block; xq: the pair copy of x (written by x's producer) -- with it, and pair weights from the fold,
        the block's launches run without an operand split and write pair copies of their own outputs for the next ones"""
        blk["x"] = x
        o1, o2, out = b[pfx + ".o1"], b[pfx + ".o2"], b[pfx + ".out"]
        c1, c2, c3, ds = blk["c1"], blk["c2"], blk["c3"], blk["ds"]
        if xq is not None and c1.wfq is not None and c2.wfq is not None and c3.wfq is not None and (ds is None or ds.wfq is not None):
            o1q, o2q, outq = b[pfx + ".o1@q"], b[pfx + ".o2@q"], b.get(pfx + ".out@q")
            K.conv_fwd(c1.geom, xq, c1.wfq, c1.bias_f, o1, relu=True, tile=c1.geom.fwd_tile_q, yq=o1q, wmeta=c1.wmeta)
            K.conv_fwd(c2.geom, o1q, c2.wfq, c2.bias_f, o2, relu=True, tile=c2.geom.fwd_tile_q, yq=o2q, wmeta=c2.wmeta)
            if ds is not None:
                idt = b[pfx + ".idt"]
                K.conv_fwd(ds.geom, xq, ds.wfq, ds.bias_f, idt, tile=ds.geom.fwd_tile_q)
            else:
                idt = x
            K.conv_fwd(c3.geom, o2q, c3.wfq, c3.bias_f, out, addend=idt, relu=True, tile=c3.geom.fwd_tile_q, yq=outq,
                       wmeta=c3.wmeta)
            return out
        if blk["po"]:       # o1 only as plane pairs (written by conv1's epilogue), conv2 without an operand split
            K.conv_fwd(c1.geom, x, c1.wf, c1.bias_f, None, relu=True, yq=o1, wmeta=c1.wmeta)
            K.conv_fwd(c2.geom, o1, c2.wf, c2.bias_f, o2, relu=True, tile=c2.geom.fwd_tile_q)
        else:
            K.conv_fwd(blk["c1"].geom, x, blk["c1"].wf, blk["c1"].bias_f, o1, relu=True)
            K.conv_fwd(blk["c2"].geom, o1, blk["c2"].wf, blk["c2"].bias_f, o2, relu=True)
        if blk["ds"] is not None:
            idt = b[pfx + ".idt"]
            K.conv_fwd(blk["ds"].geom, x, blk["ds"].wf, blk["ds"].bias_f, idt)
        else:
            idt = x
        outq = b.get(pfx + ".out@q")       # (the last block before the stages that run on pairs writes the copy they start from)
        K.conv_fwd(blk["c3"].geom, o2, blk["c3"].wf, blk["c3"].bias_f, out, addend=idt, relu=True, yq=outq,
                   wmeta=c3.wmeta if outq is not None else None)
        return out

    def _n_frozen_stages(self):
        n = 0
        while n < len(self.stages) and not self.stages[n][0]["train"]:
            n += 1
        return n

    def _prefix_forward(self, img, which):
        """Stem -> max-pool -> frozen stages of `img` into buffer set `which`, on the current stream.  Returns the set."""
        B, H, W = self.B, self.H, self.W
        b = self._pfx_sets[which]
        if self.h2:
            K.fill_zero(self.amax_pfx[which])     # every producer of this pass raises its buffer's slot from zero
        K.STAGE = "stem"
        K.stem(img, self.stem.wf, self.stem.bias_f, b["stem"], B, H, W)
        K.maxpool(b["stem"], b["pool"], B, self.stem_hw[0], self.stem_hw[1], 64, yq=b.get("pool@q"))
        x, xq = b["pool"], b.get("pool@q")
        for li in range(self._n_frozen_stages()):
            K.STAGE = f"layer{li + 1}"
            for bi, blk in enumerate(self.stages[li]):
                x = self._block_forward(blk, f"l{li + 1}.{bi}", x, b, xq)
                xq = b.get(f"l{li + 1}.{bi}.out@q")
        return b

    @staticmethod
    def _img_key(img, geo):
        return (img.data_ptr(), img._version, tuple(img.shape), geo)

    def prefetch_prefix(self, next_img):
        """Run the frozen prefix of the NEXT step's batch now, on the tower-chain stream (idle once the head's backward pass is
        through), next to the rest of this step's backward pass: the frozen stem / stages do not depend on the parameters this
        step updates (resnet.py:572-588, frozen_stages).  The next backbone_forward(next_img) picks the result up if it is called
        with the same tensor (same storage, same version counter, same geometry) and computes the prefix itself otherwise --
        bit-identical either way.  No-op when nothing is frozen, without streams, or for another geometry."""
        if self.stem.trainable or not self.use_streams or next_img is None or self.geo_key is None:
            return False
        if tuple(next_img.shape) != (self.B, 3, self.H, self.W) or not next_img.is_contiguous():
            return False
        other = 1 - self._pfx_active
        if self._pfx_sets[other] is None:                      # second buffer set + slot arena, on first use
            self._pfx_sets[other] = {}
            if self.h2:
                self.amax_pfx[other] = K.new_amax(self.dev, 128)
            for i, (name, (rows, ch, dt, twin)) in enumerate(self._pfx_shapes.items()):
                if dt == "pairs-only":
                    q = K.Planes(rows, ch, device=self.dev, kind="h2")
                    q.true_amax = self.amax_pfx[other][i]
                    self._pfx_sets[other][name] = q
                    continue
                t = torch.empty(rows, ch, device=self.dev, dtype=dt)
                self._pfx_sets[other][name] = t
                if self.h2 and dt == torch.float32:
                    self._amax_keys.append(K.register_amax(t, self.amax_pfx[other][i], by_storage=True))
                if twin:
                    q = K.Planes(rows, ch, device=self.dev, kind="h2")
                    q.true_amax = self.amax_pfx[other][i]
                    self._pfx_sets[other][name + "@q"] = q
        cs = self._chain_stream()
        self._fork(cs)                                         # (the frozen convs' folded weights are complete on this stream)
        stage = K.STAGE
        if self._pfx_event is None:
            self._pfx_event = torch.cuda.Event()   # (its own event: one of the ring's could be re-recorded by a long step)
        ev = self._pfx_event
        with torch.cuda.stream(cs):
            self._prefix_forward(next_img, other)
            K.ev_record(ev)
        K.STAGE = stage
        # the hand-over is keyed by the tensor OBJECT (kept alive here) as well as by its address / version / geometry: a
        # freed tensor's address may be handed to a new version-0 tensor by the caching allocator
        self._pfx_ready = (self._img_key(next_img, self.geo_key), other, ev, next_img)
        return True

    _pfx_event = None

    def backbone_forward(self, img):
        b = self.buf
        self._img = img if self.stem.trainable else None      # (the stem's weight gradient reads the image again)
        if self.h2:
            K.fill_zero(self.amax_act)            # every producer of this pass raises its buffer's slot from zero
        nf = self._n_frozen_stages() if not self.stem.trainable else 0
        rdy, self._pfx_ready = self._pfx_ready, None
        if not self.stem.trainable and rdy is not None and rdy[3] is img and rdy[0] == self._img_key(img, self.geo_key):
            K.ev_wait(rdy[2])                                 # prefetched during the previous step's backward pass
            self._pfx_active = rdy[1]
            pb = self._pfx_sets[self._pfx_active]
        elif not self.stem.trainable:
            pb = self._prefix_forward(img, self._pfx_active)
        else:
            K.STAGE = "stem"
            K.stem(img, self.stem.wf, self.stem.bias_f, b["stem"], self.B, self.H, self.W)
            K.maxpool(b["stem"], b["pool"], self.B, self.stem_hw[0], self.stem_hw[1], 64, yq=b.get("pool@q"))
            pb = b
        if not self.stem.trainable:
            b.update(pb)                     # the names of the prefix buffers resolve to the set this step uses
        last = "pool" if nf == 0 else f"l{nf}.{len(self.stages[nf - 1]) - 1}.out"
        x, xq = pb[last], pb.get(last + "@q")
        outs = [pb[f"l{li + 1}.{len(self.stages[li]) - 1}.out"] for li in range(nf)]
        for li in range(nf, len(self.stages)):
            blocks = self.stages[li]
            K.STAGE = f"layer{li + 1}"
            if blocks[0]["train"]:
                self._await_fold()            # trainable weights are being folded on the side stream
            for bi, blk in enumerate(blocks):
                x = self._block_forward(blk, f"l{li + 1}.{bi}", x, b, xq)
                xq = b.get(f"l{li + 1}.{bi}.out@q")
            outs.append(x)
        return outs  # C2..C5 row buffers


    def neck_forward(self, feats):
        K.STAGE = "neck"
        self._await_fold()
        b, B = self.buf, self.B
        c = feats[1:]
        for i in range(3):
            cq = b.get(f"l{i + 2}.{len(self.stages[i + 1]) - 1}.out@q")
            if cq is not None and self.lat[i].wfq is not None:      # the block's epilogue wrote C3..C5 as plane pairs too
                K.conv_fwd(self.lat[i].geom, cq, self.lat[i].wfq, self.lat[i].bias_f, b[f"lat{i}"], tile=self.lat[i].geom.fwd_tile_q)
            else:
                K.conv_fwd(self.lat[i].geom, c[i], self.lat[i].wf, self.lat[i].bias_f, b[f"lat{i}"])
        hw = self.plv.hw
        for i in (2, 1):
            K.upsample_add(b[f"lat{i - 1}"], b[f"lat{i}"], B, hw[i - 1][0], hw[i - 1][1], hw[i][0], hw[i][1], self.feat)
        P = b["P"]
        for i in range(3):
            r0, r1 = self.plv.level_rows(i)
            K.conv_fwd(self.fpn[i].geom, b[f"lat{i}"], self.fpn[i].wf, self.fpn[i].bias_f, P[r0:r1])
        for i in (3, 4):
            s0, s1 = self.plv.level_rows(i - 1)
            r0, r1 = self.plv.level_rows(i)
            K.conv_fwd(self.fpn[i].geom, P[s0:s1], self.fpn[i].wf, self.fpn[i].bias_f, P[r0:r1])
        return P

    # ------------------------------------------------------------------ two-stream helpers
    # The cls and reg towers are independent chains between P and the loss: they run on two HIP streams so
    # that the second chain fills the tails / barrier bubbles of the first (measured +10-14 % on the tower
    # GEMM pair, tools/bench_streams.py).  Weight-gradient GEMMs are likewise issued on the side stream.
    use_streams = True

    # The three extra streams are shared by every engine of the process on a device (one detector for training and another
    # one for validation must not add up to more than four streams: "stream budget" in __init__)
    _SHARED_STREAMS = {}

    def _shared_stream(self, role):
        key = (torch.device(self.dev).index if torch.device(self.dev).index is not None else torch.cuda.current_device(), role)
        st = Engine._SHARED_STREAMS.get(key)
        if st is None:
            mask = os.environ.get("RADET_WGRAD_CU_MASK")          # experiment: 32-bit pattern, repeated over the device's CUs
            if mask and role in ("side", "side2"):
                words = (C.c_uint32 * 8)(*([int(mask, 0) & 0xFFFFFFFF] * 8))
                out = C.c_void_p()
                with torch.cuda.device(self.dev):
                    _lib.check(_lib.load().radet_stream_create_cumask(words, 8, C.byref(out)), "radet_stream_create_cumask")
                st = torch.cuda.ExternalStream(out.value, device=self.dev)
            else:
                st = torch.cuda.Stream(device=self.dev)
            Engine._SHARED_STREAMS[key] = st
        return st

    def caller_stream(self):
        """A HIP stream for the CALLER's own asynchronous work next to the step (a data loader's upload stream, say) that does not
        land on the main stream's pipe.  HIP hands out hardware queues in the order of the streams' first use, and queues k and
        k + 4 share a pipe of the command processor, on which a queue waiting in a cross-stream barrier holds up the other one
        (tools/micro/stream_cliff.hip, profiles/round5_stream_cliff.txt): the engine's four streams take queues 0-3, so the
        next stream anybody uses gets queue 4 = the pipe of queue 0, the main stream -- the critical path.  This touches the
        engine's streams (so that they hold 0-3), two placeholders (queues 4, 5; kept alive) and returns the stream that got
        queue 6.  Measured (tools/bench_user_stream.py): a 14.7 MB upload per step costs 9.05 -> 9.30 ms on a plainly created
        fifth stream and 9.05 -> 9.05 ms on this one.  Call it after the first step or prepare(), before creating other streams."""
        key = (torch.device(self.dev).index if torch.device(self.dev).index is not None else torch.cuda.current_device(), "caller")
        st = Engine._SHARED_STREAMS.get(key)
        if st is None:
            touch = [self._side(), self._side2(), self._chain_stream()]
            touch += [torch.cuda.Stream(device=self.dev) for _ in range(2)]
            Engine._SHARED_STREAMS[key + ("placeholders",)] = touch[3:]
            st = Engine._SHARED_STREAMS[key] = torch.cuda.Stream(device=self.dev)
            for t in touch + [st]:
                with torch.cuda.stream(t):
                    torch.zeros(1, device=self.dev)
            torch.cuda.synchronize(self.dev)
        return st

    def _side(self):
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = self._shared_stream("side")
            # ring of reusable events; an event handed out here is consumed (waited on) at most a few layers later, so
            # the ring only has to be longer than the events of ONE step (R101: ~2 per conv + forks/joins, < 600)
            self._events = [torch.cuda.Event() for _ in range(2048)]
            self._ev_i = 0
        return self._side_stream

    def _chain_stream(self):
        if getattr(self, "_chain", None) is None:
            self._side()
            self._chain = self._shared_stream("chain")
        return self._chain

    def _event(self):
        self._ev_i = (self._ev_i + 1) % len(self._events)
        return self._events[self._ev_i]

    def _fork(self, side):
        """side stream waits for everything enqueued so far on the current stream"""
        ev = self._event()
        K.ev_record(ev)
        K.ev_wait(ev, side)

    def _join(self, side):
        ev = self._event()
        K.ev_record(ev, side)
        K.ev_wait(ev)

    # Backbone / neck weight gradients on fp16 plane pairs (round 5): dy and x are split into pairs ONCE per tensor by
    # radet_split_pairs on the weight-gradient stream (off the dgrad chain, into a scratch pair of that stream), and the GEMM
    # runs on the pair kernels (transposing LDS reads, no operand work in the loop) instead of splitting every fragment in
    # registers.  RADET_WGRAD_PAIRS=0: the in-register one-tap kernels.
    # Measured (round 5, r50 640x480 bs 4): the pair kernels take 0.64 ms less kernel time per step than the in-register ones
    # (3.19 against 3.83 ms over the 49 backbone / neck launches), the 98 split launches give it back: 8.94 against 8.84 ms per
    # step.  Off by default; the kernels stay (they are what a producer-side pair copy would feed, DESIGN.md "next").
    wgrad_pairs = os.environ.get("RADET_WGRAD_PAIRS", "0") == "1"

    def _wgrad_pairs_ok(self, geom, dy, x):
        return (self.h2 and self.wgrad_pairs and not K._isp(dy) and geom.cin % 32 == 0 and geom.cout % 32 == 0
                and dy.dtype == torch.float32 and dy.dim() == 2 and dy.shape[1] == geom.cout and x.shape[1] == geom.cin)

    def _wgrad_via_pairs(self, geom, dy, x, slabs, dbias_partials):
        """on the current (weight-gradient) stream: dy, x -> this stream's scratch pairs -> pair weight-gradient GEMM"""
        key = torch.cuda.current_stream().cuda_stream
        sc = self._pair_scratch.get(key)
        need = max(dy.numel(), x.numel())
        if sc is None or sc[0].numel() < 2 * need:
            n = max(2 * need, 2 * self._pair_scratch_elems())
            sc = self._pair_scratch[key] = (torch.empty(n, dtype=torch.float16, device=self.dev),
                                            torch.empty(n, dtype=torch.float16, device=self.dev), K.new_amax(self.dev, 2))
        dyq = K.Planes(dy.shape[0], geom.cout, t=sc[0][:2 * dy.numel()].view(dy.shape[0], 2 * geom.cout), kind="h2", amax=sc[2][0])
        xq = K.Planes(x.shape[0], geom.cin, t=sc[1][:2 * x.numel()].view(x.shape[0], 2 * geom.cin), kind="h2", amax=sc[2][1])
        K.split_planes(dy, dyq)
        K.split_planes(x, xq)
        K.conv_wgrad(geom, dyq, xq, slabs, dbias_partials)

    def _pair_scratch_elems(self):
        return max((max(c.geom.lout.rows * c.cout, c.geom.lin.rows * c.cin) for c in self.convs
                    if c.trainable and c.geom is not None and hasattr(c.geom, "lout")), default=0)

    _pair_scratch = None

    def _wgrad_async(self, geom, dy, x, slabs, dbias_partials=None, conv=None):
        """Weight-gradient GEMM off the critical path: issued on the side stream once `dy` is ready (or collected
        for the stage's grouped launch, see flush_wgrads)."""
        if conv is not None and conv.grouped:
            self._pending_wgrad.append(dict(g=geom, dy=dy, x=x, slabs=slabs, dbias=dbias_partials))
            return None
        if self._pair_scratch is None:
            self._pair_scratch = {}
        wg = self._wgrad_via_pairs if self._wgrad_pairs_ok(geom, dy, x) else K.conv_wgrad
        if not self.use_streams:
            wg(geom, dy, x, slabs, dbias_partials)
            return None
        # weight-gradient GEMMs are small grids (250-800 workgroups): alternate them between the side stream and a
        # second one so that two run concurrently and fill each other's tails
        side = self._side()
        if self.wgrad_streams > 1:
            self._wg_i = getattr(self, "_wg_i", 0) + 1
            if self._wg_i & 1:
                side = self._side2()
                self._side2_dirty = True
        self._fork(side)
        with torch.cuda.stream(side):
            wg(geom, dy, x, slabs, dbias_partials)
            ev = self._event()
            K.ev_record(ev)
        return ev   # completes when this weight-gradient GEMM has finished reading dy / x

    wgrad_streams = int(os.environ.get("RADET_WGRAD_STREAMS", "2"))
    _side2_dirty = False

    def _side2(self):
        if getattr(self, "_side2_stream", None) is None:
            self._side()
            self._side2_stream = self._shared_stream("side2")
        return self._side2_stream

    def side_collect(self):
        """Side stream waits for the wgrads issued on the second wgrad stream (call on the way to a slab reduction)."""
        if self._side2_dirty:
            ev = self._event()
            K.ev_record(ev, self._side2_stream)
            K.ev_wait(ev, self._side())
            self._side2_dirty = False

    def join_side(self):
        """Current stream waits for all side-stream work (call before consuming wgrad slabs)."""
        if self.use_streams and getattr(self, "_side_stream", None) is not None:
            self.side_collect()
            self._join(self._side_stream)

    # GroupNorm + ReLU of the towers on fp32 tensors (every path but the default plane-pair one: RADET_P3=0, RADET_TOWER_MODE).
    # With the fp16 hi / lo arithmetic the tensors they write are conv operands and need their amax slots raised: the forward
    # kernel's _q variant does it for y, the backward's dz gets a stand-alone pass (these are comparison modes, not the default).
    def _gn_fwd(self, z, gamma, beta, y, stats, ws):
        if self.h2 and not K._isp(y):
            K.gn_relu_fwd_q(self.plv, z, gamma, beta, y, None, stats, ws)
        else:
            K.gn_relu_fwd(self.plv, z, gamma, beta, y, stats, ws)

    def _gn_bwd(self, dy, z, stats, gamma, beta, dz, dgamma, dbeta, ws):
        K.gn_relu_bwd(self.plv, dy, z, stats, gamma, beta, dz, dgamma, dbeta, ws)
        if self.h2 and not K._isp(dz) and dz.dtype == torch.float32:
            sl = K.amax_slot(dz)
            if sl is not None:
                K.absmax(dz, sl)

    def _tower_fwd_layer(self, t, tower, i, x, ws):
        b, p = self.buf, self.p
        c = tower[i]
        z, y = b[f"{t}.z{i}"], b[f"{t}.y{i}"]
        self._tower_launch(K.conv_fwd, c.geom, x, c.wf, None, z, tile=self._ttile(c))
        gn = f"bbox_head.{t}_convs.{i}.gn"
        self._gn_fwd(z, p[gn + ".weight"], p[gn + ".bias"], y, b[f"{t}.stats{i}"], ws)
        return y

    tower_fwd_streams = os.environ.get("RADET_TOWER_FWD_STREAMS", "1") != "0"

    def _tower_fwd_layer_p(self, t, tower, i, x, ws):
        """one tower layer with plane operands on the current stream: GEMM (256 x 128 tiles, no split-K) + GroupNorm + ReLU"""
        b, p = self.buf, self.p
        c = tower[i]
        z, y = b[f"{t}.z{i}"], b[f"{t}.y{i}"]
        self._tower_launch(K.conv_fwd, c.geom, x, c.wf, None, z, tile=self.tower_tile | 0x100 | (1 << 12) | self.tower_stages)
        gn = f"bbox_head.{t}_convs.{i}.gn"
        pl = K._isp(y)                  # the last layer's output feeds the predictor convs: fp32
        if self.h2:
            K.gn_relu_fwd_q(self.plv, z, p[gn + ".weight"], p[gn + ".bias"], None if pl else y, y if pl else None,
                            b[f"{t}.stats{i}"], ws, zhat_amax=b[f"{t}.zhat{i}"])
            return y
        K.gn_relu_fwd_p(self.plv, z, p[gn + ".weight"], p[gn + ".bias"], None if pl else y, y if pl else None,
                        b[f"{t}.stats{i}"], ws)
        return y

    # "streams": cls / reg towers on two HIP streams; "pair": cls+reg layer = one grouped launch (forward and backward);
    # "hybrid" (default): grouped forward launches (they run alone on the device -> clean roofline measurement),
    # two-stream backward (dgrad / wgrad / GroupNorm of the two towers overlap).  bench: 229.5 / 227.7 / 232 img/s
    # "pairbwd" (default with the bf16-plane fp32 arithmetic): grouped launches forward AND for the tower dgrads (their
    # 128 x 128 tiles fill the chip better two GEMMs at a time: -1 % step), wgrads on the side streams, only the forward
    # launches tagged; with the native fp32 MFMA the two-stream backward of "hybrid" is the faster one (14.27 vs 14.34 ms)
    tower_mode = os.environ.get("RADET_TOWER_MODE", "hybrid")

    def _tower_pair_fwd(self, i, xc, xr):
        """cls_convs[i] and reg_convs[i] as ONE grouped GEMM launch, then the two GroupNorm+ReLU."""
        b, p = self.buf, self.p
        cc, cr = self.cls_tower[i], self.reg_tower[i]
        zc, yc, zr, yr = b[f"cls.z{i}"], b[f"cls.y{i}"], b[f"reg.z{i}"], b[f"reg.y{i}"]
        self._tower_launch(K.conv_fwd_pair, cc.geom, dict(x=xc, w=cc.wf, y=zc), dict(x=xr, w=cr.wf, y=zr),
                           tile=self._ttile(cc))
        gc, gr = f"bbox_head.cls_convs.{i}.gn", f"bbox_head.reg_convs.{i}.gn"
        if self.p3:
            pl = K._isp(yc)             # the last layer's output feeds the predictor convs: fp32
            if self.h2:
                K.gn_relu_fwd_pair_q(self.plv, (zc, p[gc + ".weight"], p[gc + ".bias"], None if pl else yc, yc if pl else None,
                                                b[f"cls.stats{i}"], self.gn_ws, b[f"cls.zhat{i}"]),
                                     (zr, p[gr + ".weight"], p[gr + ".bias"], None if pl else yr, yr if pl else None,
                                      b[f"reg.stats{i}"], self.gn_ws2, b[f"reg.zhat{i}"]))
                return yc, yr
            K.gn_relu_fwd_pair_p(self.plv, (zc, p[gc + ".weight"], p[gc + ".bias"], None if pl else yc, yc if pl else None,
                                            b[f"cls.stats{i}"], self.gn_ws),
                                 (zr, p[gr + ".weight"], p[gr + ".bias"], None if pl else yr, yr if pl else None,
                                  b[f"reg.stats{i}"], self.gn_ws2))
            return yc, yr
        # both GroupNorms in one pair of launches (on two streams the fork and the join idled the device for longer
        # than the 23 us of kernels they overlapped)
        if self.h2:       # (fp32 y: the _q kernel raises its amax slot)
            K.gn_relu_fwd_pair_q(self.plv, (zc, p[gc + ".weight"], p[gc + ".bias"], yc, None, b[f"cls.stats{i}"], self.gn_ws, None),
                                 (zr, p[gr + ".weight"], p[gr + ".bias"], yr, None, b[f"reg.stats{i}"], self.gn_ws2, None))
            return yc, yr
        K.gn_relu_fwd_pair(self.plv, (zc, p[gc + ".weight"], p[gc + ".bias"], yc, b[f"cls.stats{i}"], self.gn_ws),
                           (zr, p[gr + ".weight"], p[gr + ".bias"], yr, b[f"reg.stats{i}"], self.gn_ws2))
        return yc, yr

    def head_forward(self, P):
        K.STAGE = "head"
        b = self.buf
        n = self.stacked_convs
        if self.tower_mode in ("pair", "hybrid", "pairbwd"):
            xc = xr = P
            if self.p3:
                K.split_planes(P, b["Pp"])
                xc = xr = b["Pp"]
            if self.p3 and self.tower_fwd_streams and self.use_streams:
                # plane operands: a tower GEMM is 200 tiles of 256 x 128 and a tile owns a CU, so the grouped cls + reg launch
                # is two rounds on 256 CUs with the second 56 % full.  The two towers as two chains on two streams instead:
                # the tiles of one chain's GEMM take the CUs the other chain's GEMM leaves free (1600 tiles = 6.25 rounds
                # instead of 8), its GroupNorm runs next to the other chain's GEMM
                side = self._side()
                self._fork(side)
                for i in range(n):
                    xc = self._tower_fwd_layer_p("cls", self.cls_tower, i, xc, self.gn_ws)
                    with torch.cuda.stream(side):
                        xr = self._tower_fwd_layer_p("reg", self.reg_tower, i, xr, self.gn_ws2)
            else:
                side = None
                for i in range(n):
                    xc, xr = self._tower_pair_fwd(i, xc, xr)
            reg_stream = torch.cuda.stream(side) if side is not None else contextlib.nullcontext()
            if self.x3 and self.feat % 16 == 0 and self.pred_cls.cout <= 32 and self.pred_reg.cout + self.pred_iou.cout <= 32 \
                    and os.environ.get("RADET_PRED_PATCH", "1") != "0":      # (the patch kernel's tile has 32 output columns)
                # direct convolution from an LDS patch: the tower output is fetched 1.4 times instead of 9, reg + iou
                # share one launch
                pc, pr, pi = self.pred_cls, self.pred_reg, self.pred_iou
                K.pred_conv_patch(self.plv, xc, (pc.wf, pc.bias_f, b["cls"], pc.cout))
                with reg_stream:
                    K.pred_conv_patch(self.plv, xr, (pr.wf, pr.bias_f, b["reg_u"], pr.cout), (pi.wf, pi.bias_f, b["iou"], pi.cout))
            else:
                K.conv_fwd(self.pred_cls.geom, xc, self.pred_cls.wf, self.pred_cls.bias_f, b["cls"])
                with reg_stream:
                    K.conv_fwd(self.pred_reg.geom, xr, self.pred_reg.wf, self.pred_reg.bias_f, b["reg_u"])
                    K.conv_fwd(self.pred_iou.geom, xr, self.pred_iou.wf, self.pred_iou.bias_f, b["iou"])
            if side is not None:
                self._join(side)
        elif self.use_streams:
            side = self._side()
            self._fork(side)
            xc = xr = P
            for i in range(n):
                xc = self._tower_fwd_layer("cls", self.cls_tower, i, xc, self.gn_ws)
                with torch.cuda.stream(side):
                    xr = self._tower_fwd_layer("reg", self.reg_tower, i, xr, self.gn_ws2)
            K.conv_fwd(self.pred_cls.geom, xc, self.pred_cls.wf, self.pred_cls.bias_f, b["cls"])
            with torch.cuda.stream(side):
                K.conv_fwd(self.pred_reg.geom, xr, self.pred_reg.wf, self.pred_reg.bias_f, b["reg_u"])
                K.conv_fwd(self.pred_iou.geom, xr, self.pred_iou.wf, self.pred_iou.bias_f, b["iou"])
            self._join(side)
        else:
            xc = xr = P
            for i in range(n):
                xc = self._tower_fwd_layer("cls", self.cls_tower, i, xc, self.gn_ws)
            for i in range(n):
                xr = self._tower_fwd_layer("reg", self.reg_tower, i, xr, self.gn_ws)
            K.conv_fwd(self.pred_cls.geom, xc, self.pred_cls.wf, self.pred_cls.bias_f, b["cls"])
            K.conv_fwd(self.pred_reg.geom, xr, self.pred_reg.wf, self.pred_reg.bias_f, b["reg_u"])
            K.conv_fwd(self.pred_iou.geom, xr, self.pred_iou.wf, self.pred_iou.bias_f, b["iou"])
        return b["cls"], b["reg_u"], b["iou"]

    def scales_tensor(self):
        """The five learnable Scale scalars as one contiguous device vector (views a flat arena when the
        owner laid them out contiguously, else packs them)."""
        names = [f"bbox_head.scales.{i}.scale" for i in range(len(self.strides))]
        ts = [self.p[n] for n in names]
        base = ts[0].data_ptr()
        if all(t.data_ptr() == base + 4 * i for i, t in enumerate(ts)):
            return torch.as_strided(ts[0], (len(ts),), (1,))
        return torch.stack([t.reshape(()) for t in ts]).contiguous()

    def loss(self, gt_boxes, gt_labels, gt_off, p2g, pw, grad_scale=None, alpha=0.25, gamma=2.0, lbw=2.0,
             labels_out=None, tgt_out=None):
        b = self.buf
        dri = b["dregiou"]
        K.head_loss(b["cls"], b["reg_u"], b["iou"], self.scales_tensor(), gt_boxes, gt_labels, gt_off, p2g, pw, self.ldesc,
                    self.nlvl, self.B, self.num_classes, alpha, gamma, lbw, 1e-6, grad_scale, self.losses, b["dcls"],
                    self.cls_pad, dri, 16, dri.view(-1)[4:], 16, self.dscales, self.loss_ws, labels_out, tgt_out)
        if self.h2:       # the predictors' dgrad / wgrad scale their dy operand by its largest magnitude
            K.absmax(b["dcls"], K.amax_slot(b["dcls"]))
            K.absmax(dri, K.amax_slot(dri))
        if self.h16:      # the predictors' dgrad / wgrad read bf16: reg -> cols 0-3, iou -> col 8 (16-byte aligned)
            K.convert_rows(b["dcls"], b["dcls16"])
            K.convert_rows(dri, b["dregiou16"], ncols=4)
            K.convert_rows(dri, b["dregiou16"], ncols=1, src_off=4, dst_off=self.iou_col)
        return self.losses

    def _head_grads(self):
        b = self.buf
        return (b["dcls16"], b["dregiou16"]) if self.h16 else (b["dcls"], b["dregiou"])

    # ------------------------------------------------------------------ backward
    def _tower_bwd_head(self, t):
        """predictor wgrad + dgrad into the tower's dy buffer"""
        b = self.buf
        ylast = b[f"{t}.y{self.stacked_convs - 1}"]
        dy = b[f"{t}.dy"]
        dcls, dri = self._head_grads()
        if t == "cls":
            pc = self.pred_cls
            K.conv_wgrad(pc.geom, dcls, ylast, pc.slabs, pc.dbias_partials, cout=pc.cout, ld_dy=self.cls_pad)
            K.conv_dgrad(pc.geom, dcls, pc.wft, dy, k_channels=self.cls_pad)
        else:
            pr, pi = self.pred_reg, self.pred_iou
            K.conv_wgrad(pr.geom, dri, ylast, pr.slabs, pr.dbias_partials, cout=4, ld_dy=self.ri_pad)
            K.conv_wgrad(pi.geom, dri.view(-1)[self.iou_col:], ylast, pi.slabs, pi.dbias_partials, cout=1, ld_dy=self.ri_pad)
            K.conv_dgrad(pr.geom, dri, pr.wft, dy, k_channels=self.ri_pad)

    def _tower_bwd_head_async(self, t):
        """like _tower_bwd_head, with the predictor weight-gradient GEMMs on the side stream"""
        b = self.buf
        ylast = b[f"{t}.y{self.stacked_convs - 1}"]
        dy = b[f"{t}.dy"]
        side = self._side() if self.use_streams else None

        def wg(*a, **k):
            if side is None:
                K.conv_wgrad(*a, **k)
            else:
                self._fork(side)
                with torch.cuda.stream(side):
                    K.conv_wgrad(*a, **k)

        dcls, dri = self._head_grads()
        if t == "cls":
            pc = self.pred_cls
            wg(pc.geom, dcls, ylast, pc.slabs, pc.dbias_partials, cout=pc.cout, ld_dy=self.cls_pad)
            K.conv_dgrad(pc.geom, dcls, pc.wft, dy, k_channels=self.cls_pad)
        else:
            pr, pi = self.pred_reg, self.pred_iou
            wg(pr.geom, dri, ylast, pr.slabs, pr.dbias_partials, cout=4, ld_dy=self.ri_pad)
            wg(pi.geom, dri.view(-1)[self.iou_col:], ylast, pi.slabs, pi.dbias_partials, cout=1, ld_dy=self.ri_pad)
            K.conv_dgrad(pr.geom, dri, pr.wft, dy, k_channels=self.ri_pad)

    def _tower_bwd_layer(self, t, tower, i, ws, dP, addend):
        b, p, g = self.buf, self.p, self.g
        c = tower[i]
        dy, dz = b[f"{t}.dy"], b[f"{t}.dz"]
        gn = f"bbox_head.{t}_convs.{i}.gn"
        self._gn_bwd(dy, b[f"{t}.z{i}"], b[f"{t}.stats{i}"], p[gn + ".weight"], p[gn + ".bias"], dz,
                     g[gn + ".weight"], g[gn + ".bias"], ws)
        x = b[f"{t}.y{i - 1}"] if i > 0 else b["P"]
        # the weight-gradient GEMM stays ON the tower's chain: moved to a stream of its own (overlapping the other
        # tower's dgrad and GroupNorm) the all-taps kernel's large workgroups starve the dependent chain -- measured
        # 20.4 instead of 16.0 ms per step
        K.conv_wgrad(c.geom, dz, x, c.slabs, None)
        if self.tower_mode == "hybrid":   # only the (un-overlapped) forward launches carry the profiling tag
            K.conv_dgrad(c.geom, dz, c.wft, dy if i > 0 else dP, addend=None if i > 0 else addend,
                         tile=self._ttile(c, bwd=True, tag=False))
        elif i > 0:
            self._tower_launch(K.conv_dgrad, c.geom, dz, c.wft, dy, tile=self._ttile(c, bwd=True))
        else:
            self._tower_launch(K.conv_dgrad, c.geom, dz, c.wft, dP, addend=addend, tile=self._ttile(c, bwd=True))

    def head_backward(self):
        """Consumes buf['dcls'] / buf['dregiou'] (written by loss()); leaves dL/dP in buf['dP']."""
        K.STAGE = "head"
        b = self.buf
        n = self.stacked_convs
        dP = b["dP"]
        if self.p3 and self.tower_mode == "pairbwd" and self.use_streams and os.environ.get("RADET_TOWER_BWD_CHAINS", "1") != "0":
            # plane operands: the two towers' backward as two CHAINS (cls on this stream, reg on the chain stream), each layer
            # GroupNorm backward -> weight gradient (on the side streams, as before) -> dgrad as ONE 200-tile launch.  The
            # grouped cls + reg dgrad (400 tiles of 256 x 128 that own a CU each = two rounds, the second 56 % full) held
            # every CU while both chains' next GroupNorms waited for it; as single launches the tiles of one chain's dgrad, the
            # other's and the weight gradients interleave CU by CU: step -2.5 % (same box, same run)
            p, g = self.p, self.g
            cs = self._chain_stream()
            wg_done = {}

            def layer(t, tower, ws, i):
                gn = f"bbox_head.{t}_convs.{i}.gn"
                ev = wg_done.get((t, i + 2))          # dz[i & 1] was last read by the wgrad of layer i + 2
                if ev is not None:
                    K.ev_wait(ev)
                if self.h2:
                    K.gn_relu_bwd_q(self.plv, b[f"{t}.dy"], b[f"{t}.z{i}"], b[f"{t}.stats{i}"], p[gn + ".weight"],
                                    p[gn + ".bias"], None, b[f"{t}.dz{i & 1}"], g[gn + ".weight"], g[gn + ".bias"], ws,
                                    zhat_amax=b[f"{t}.zhat{i}"])
                else:
                    K.gn_relu_bwd_p(self.plv, b[f"{t}.dy"], b[f"{t}.z{i}"], b[f"{t}.stats{i}"], p[gn + ".weight"],
                                    p[gn + ".bias"], None, b[f"{t}.dz{i & 1}"], g[gn + ".weight"], g[gn + ".bias"], ws)
                x = b[f"{t}.y{i - 1}"] if i > 0 else b["Pp"]
                wg_done[(t, i)] = self._wgrad_async(tower[i].geom, b[f"{t}.dz{i & 1}"], x, tower[i].slabs, None)
                if i > 0:
                    K.conv_dgrad(tower[i].geom, b[f"{t}.dz{i & 1}"], tower[i].wft, b[f"{t}.dy"], tile=self.tower_tile | (1 << 12))
            self._tower_bwd_head_async("cls")
            self._fork(cs)
            with torch.cuda.stream(cs):
                self._tower_bwd_head_async("reg")
            for i in range(n - 1, -1, -1):
                layer("cls", self.cls_tower, self.gn_ws, i)
                with torch.cuda.stream(cs):
                    layer("reg", self.reg_tower, self.gn_ws2, i)
            # both first layers write dL/dP: cls on this stream, then reg accumulates onto it on the chain stream
            K.conv_dgrad(self.cls_tower[0].geom, b["cls.dz0"], self.cls_tower[0].wft, dP, tile=self.tower_tile | (1 << 12))
            self._fork(cs)
            with torch.cuda.stream(cs):
                K.conv_dgrad(self.reg_tower[0].geom, b["reg.dz0"], self.reg_tower[0].wft, dP, addend=dP, tile=self.tower_tile | (1 << 12))
            self._join(cs)
        elif self.tower_mode in ("pair", "pairbwd"):
            p, g = self.p, self.g
            tagged = self.tower_mode == "pair"        # "pairbwd": only the forward launches are tagged / timed (like hybrid)
            launch = self._tower_launch if tagged else (lambda fn, *a, **k: fn(*a, **k))
            self._tower_bwd_head_async("cls")
            self._tower_bwd_head_async("reg")
            wg_done = {}
            # plane-operand GEMMs own a CU's LDS per workgroup: a weight-gradient launch already running starves the other
            # tower's GroupNorm backward of dispatch slots (20 -> 190 us per layer on the dependent chain), so with them both
            # GroupNorms go first and the two weight-gradient GEMMs follow
            gn_first = self.p3 and os.environ.get("RADET_GN_FIRST", "1") != "0"
            for i in range(n - 1, -1, -1):
                pending = []
                for t, tower in (("cls", self.cls_tower), ("reg", self.reg_tower)):
                    gn = f"bbox_head.{t}_convs.{i}.gn"
                    ev = wg_done.get((t, i + 2))          # dz[i & 1] was last read by the wgrad of layer i + 2
                    if ev is not None:
                        K.ev_wait(ev)
                    if self.p3 and self.h2:
                        K.gn_relu_bwd_q(self.plv, b[f"{t}.dy"], b[f"{t}.z{i}"], b[f"{t}.stats{i}"], p[gn + ".weight"],
                                        p[gn + ".bias"], None, b[f"{t}.dz{i & 1}"], g[gn + ".weight"], g[gn + ".bias"], self.gn_ws,
                                        zhat_amax=b[f"{t}.zhat{i}"])
                    elif self.p3:
                        K.gn_relu_bwd_p(self.plv, b[f"{t}.dy"], b[f"{t}.z{i}"], b[f"{t}.stats{i}"], p[gn + ".weight"],
                                        p[gn + ".bias"], None, b[f"{t}.dz{i & 1}"], g[gn + ".weight"], g[gn + ".bias"], self.gn_ws)
                    else:
                        self._gn_bwd(b[f"{t}.dy"], b[f"{t}.z{i}"], b[f"{t}.stats{i}"], p[gn + ".weight"],
                                     p[gn + ".bias"], b[f"{t}.dz{i & 1}"], g[gn + ".weight"], g[gn + ".bias"], self.gn_ws)
                    x = b[f"{t}.y{i - 1}"] if i > 0 else (b["Pp"] if self.p3 else b["P"])
                    if gn_first:
                        pending.append((t, tower, x))
                    else:
                        wg_done[(t, i)] = self._wgrad_async(tower[i].geom, b[f"{t}.dz{i & 1}"], x, tower[i].slabs, None)
                if pending and os.environ.get("RADET_TOWER_WGRAD_MAIN") == "1":     # experiment: no co-running in the towers
                    for t, tower, x in pending:
                        K.conv_wgrad(tower[i].geom, b[f"{t}.dz{i & 1}"], x, tower[i].slabs, None)
                    pending = []
                for t, tower, x in pending:
                    wg_done[(t, i)] = self._wgrad_async(tower[i].geom, b[f"{t}.dz{i & 1}"], x, tower[i].slabs, None)
                cc, cr = self.cls_tower[i], self.reg_tower[i]
                dzc, dzr = b[f"cls.dz{i & 1}"], b[f"reg.dz{i & 1}"]
                if i > 0:
                    launch(K.conv_dgrad_pair, cc.geom, dict(x=dzc, w=cc.wft, y=b["cls.dy"]),
                                       dict(x=dzr, w=cr.wft, y=b["reg.dy"]), tile=self._ttile(cc, bwd=True, tag=tagged))
                else:   # both write dL/dP: the second accumulates onto the first
                    launch(K.conv_dgrad, cc.geom, dzc, cc.wft, dP, tile=self._ttile(cc, bwd=True, tag=tagged, pair=False))
                    launch(K.conv_dgrad, cr.geom, dzr, cr.wft, dP, addend=dP, tile=self._ttile(cr, bwd=True, tag=tagged, pair=False))
        elif self.use_streams:
            side = self._side()
            self._fork(side)
            self._tower_bwd_head("cls")
            with torch.cuda.stream(side):
                self._tower_bwd_head("reg")
            for i in range(n - 1, 0, -1):
                self._tower_bwd_layer("cls", self.cls_tower, i, self.gn_ws, dP, None)
                with torch.cuda.stream(side):
                    self._tower_bwd_layer("reg", self.reg_tower, i, self.gn_ws2, dP, None)
            self._tower_bwd_layer("cls", self.cls_tower, 0, self.gn_ws, dP, None)     # writes dP
            self._fork(side)                                                          # reg's last dgrad adds onto it
            with torch.cuda.stream(side):
                self._tower_bwd_layer("reg", self.reg_tower, 0, self.gn_ws2, dP, dP)
            self._join(side)
        else:
            for t, tower, first in (("cls", self.cls_tower, True), ("reg", self.reg_tower, False)):
                self._tower_bwd_head(t)
                for i in range(n - 1, -1, -1):
                    self._tower_bwd_layer(t, tower, i, self.gn_ws, dP, None if first else dP)
        self._write_scale_grads()
        return dP

    def _write_scale_grads(self):
        g = self.g
        ts = [g[f"bbox_head.scales.{i}.scale"] for i in range(len(self.strides))]
        base = ts[0].data_ptr()
        if all(t.data_ptr() == base + 4 * i for i, t in enumerate(ts)):      # contiguous in the gradient arena: one copy
            K.copy_d2d(torch.as_strided(ts[0], (len(ts),), (1,)), self.dscales)
            return
        for i, t in enumerate(ts):
            K.copy_d2d(t, self.dscales[i])

    def neck_backward(self, dP):
        K.STAGE = "neck"
        b, B, f = self.buf, self.B, self.feat
        P = b["P"]
        lr = [self.plv.level_rows(i) for i in range(5)]
        sl = lambda t, i: t[lr[i][0]:lr[i][1]]  # noqa: E731
        tmp2, tmp3 = b["dP_tmp2"], b["dP_tmp3"]
        # P7 = conv4(P6); P6 = conv3(P5)
        c4, c3 = self.fpn[4], self.fpn[3]
        self._wgrad_async(c4.geom, sl(dP, 4), sl(P, 3), c4.slabs, c4.dbias_partials, conv=c4)
        K.conv_dgrad(c4.geom, sl(dP, 4), c4.wft, tmp3, addend=sl(dP, 3))
        self._wgrad_async(c3.geom, tmp3, sl(P, 2), c3.slabs, c3.dbias_partials, conv=c3)
        K.conv_dgrad(c3.geom, tmp3, c3.wft, tmp2, addend=sl(dP, 2))
        srcs = [sl(dP, 0), sl(dP, 1), tmp2]
        for i in range(3):
            c = self.fpn[i]
            self._wgrad_async(c.geom, srcs[i], b[f"lat{i}"], c.slabs, c.dbias_partials, conv=c)
            K.conv_dgrad(c.geom, srcs[i], c.wft, b[f"d_lat{i}"])
        hw = self.plv.hw
        for i in (1, 2):
            K.upsample_add_bwd(b[f"d_lat{i}"], b[f"d_lat{i - 1}"], B, hw[i - 1][0], hw[i - 1][1], hw[i][0], hw[i][1], f)
        feats = [self.stages[i][-1] for i in (1, 2, 3)]
        for i in range(3):
            c = self.lat[i]
            x = b[f"l{i + 2}.{len(self.stages[i + 1]) - 1}.out"]
            self._wgrad_async(c.geom, b[f"d_lat{i}"], x, c.slabs, c.dbias_partials, conv=c)
            K.conv_dgrad(c.geom, b[f"d_lat{i}"], c.wft, b[f"d_c{i}"])
        del feats
        self.flush_wgrads()
        return [b["d_c0"], b["d_c1"], b["d_c2"]]

    def backbone_backward(self, d_feats, after_stage=None, block_ends=()):
        """d_feats: gradients w.r.t. C3, C4, C5 coming from the neck. `after_stage(li)` is called once
        all weight gradients of stage li are complete (bucketed unfold / all-reduce hook); for the (stage, block) pairs in
        `block_ends` also `after_stage(li, bi)` right after block bi's weight gradients (per-block gradient buckets)."""
        b = self.buf
        d_next_pre = None     # d_pre of the block after the current one (same stage or next stage's first block)
        nxt = None            # that block
        for li in range(len(self.stages) - 1, -1, -1):
            blocks = self.stages[li]
            if not blocks[0]["train"]:
                break
            K.STAGE = f"layer{li + 1}"        # (the dgrad into this stage's last block reads the next stage's conv1 / shortcut)
            for bi in range(len(blocks) - 1, -1, -1):
                blk = blocks[bi]
                pfx = f"l{li + 1}.{bi}"
                out, o1, o2 = b[pfx + ".out"], b[pfx + ".o1"], b[pfx + ".o2"]
                d_pre, d_o1, d_o2 = b[pfx + ".d_pre"], b[pfx + ".d_o1"], b[pfx + ".d_o2"]
                # ---- gradient w.r.t. this block's (post-ReLU) output -> d_pre = grad * [out > 0]
                last_of_stage = bi == len(blocks) - 1
                ext = d_feats[li - 1] if (last_of_stage and li >= 1) else None   # FPN taps C3..C5 (stages 2..4)
                if nxt is None:
                    # top of the network: only the FPN gradient reaches C5
                    K.relu_bwd(ext, None, out, d_pre)
                elif nxt["ds"] is not None:
                    # next block is the first of the next stage: conv1 + downsample both read `out`
                    # d_pre = [out > 0] * (dgrad_conv1 + ext + dgrad_downsample): the dense conv1 term first, then the
                    # strided 1x1 downsample accumulates in place on the quarter of the positions it reaches
                    K.conv_dgrad(nxt["c1"].geom, nxt["d_o1"], nxt["c1"].wft, d_pre, addend=ext, mask=out)
                    K.conv_dgrad(nxt["ds"].geom, nxt["d_pre"], nxt["ds"].wft, d_pre, addend=d_pre, mask=out,
                                 skip_zero_rows=True)
                else:
                    # identity shortcut: d_out = dgrad_conv1(next) + d_pre(next)
                    K.conv_dgrad(nxt["c1"].geom, nxt["d_o1"], nxt["c1"].wft, d_pre, addend=nxt["d_pre"], mask=out)
                blk["d_pre"], blk["d_o1"] = d_pre, d_o1
                # ---- inside the block
                c1, c2, c3, ds = blk["c1"], blk["c2"], blk["c3"], blk["ds"]
                self._wgrad_async(c3.geom, d_pre, o2, c3.slabs, c3.dbias_partials, conv=c3)
                if blk["po"]:     # d_o2 only as plane pairs (conv3's dgrad epilogue writes them), o1 is one: conv2's weight
                    #               gradient and dgrad run on pairs, the ReLU mask of d_o1 is read from o1's pairs
                    K.conv_dgrad(c3.geom, d_pre, c3.wft, None, mask=o2, yq=d_o2, wmeta=c3.wmeta_t)
                else:
                    K.conv_dgrad(c3.geom, d_pre, c3.wft, d_o2, mask=o2)
                self._wgrad_async(c2.geom, d_o2, o1, c2.slabs, c2.dbias_partials, conv=c2)
                K.conv_dgrad(c2.geom, d_o2, c2.wft, d_o1, mask=o1)
                self._wgrad_async(c1.geom, d_o1, blk["x"], c1.slabs, c1.dbias_partials, conv=c1)
                if ds is not None:
                    self._wgrad_async(ds.geom, d_pre, blk["x"], ds.slabs, ds.dbias_partials, conv=ds)
                nxt = blk
                if (li, bi) in block_ends and after_stage is not None:
                    self.flush_wgrads()
                    after_stage(li, bi)
                elif self.wgrad_flush_blocks and (len(blocks) - bi) % self.wgrad_flush_blocks == 0:
                    self.flush_wgrads()
            self.flush_wgrads()
            if after_stage is not None:
                after_stage(li)
        if self.stem.trainable and nxt is not None and nxt is self.stages[0][0]:
            # frozen_stages = -1: through layer1.0's conv1 + projection shortcut to the pooled map, through the max-pool and
            # the stem's ReLU, into conv1 / bn1 (resnet.py:572-588 leaves them trainable)
            d_pool, d_stem = b["d_pool"], b["d_stem"]
            K.conv_dgrad(nxt["c1"].geom, nxt["d_o1"], nxt["c1"].wft, d_pool)
            K.conv_dgrad(nxt["ds"].geom, nxt["d_pre"], nxt["ds"].wft, d_pool, addend=d_pool)
            K.maxpool_bwd_relu(b["stem"], d_pool, d_stem, self.B, self.stem_hw[0], self.stem_hw[1], 64)
            K.stem_wgrad(self._img, d_stem, self.stem.slabs, self.stem.dbias_partials, self.B, self.H, self.W,
                         self.stem.geom.nsplit)
            if after_stage is not None:
                after_stage("stem")
        return None

