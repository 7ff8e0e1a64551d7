"""Thin Python launchers over the C ABI (include/radet_hip.h): torch tensors are used only as device
buffers (data_ptr) and for the current HIP stream.  No arithmetic happens in PyTorch here."""
import ctypes as C
import os

import torch

from . import _lib


def _ptr(t):
    """device address of a contiguous device tensor (None -> NULL) as a plain int: ctypes converts it per the argtypes.
    This runs ~1500 times per train step: the bf16-storage step and single-image inference are bound by the host's enqueue
    time, so the checks are two attribute reads and one call."""
    if t is None:
        return None
    if not (t.is_cuda and t.is_contiguous()):
        raise AssertionError("HIP kernels need contiguous device tensors")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """the current HIP stream's handle (plain int).  torch._C._cuda_getCurrentRawStream avoids building a torch.cuda.Stream
    object per launch (2 us each, 250 launches per step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def ev_record(ev, stream=None):
    """ev.record() on `stream` (a torch stream; None: the current one).  The engine's cross-stream dependencies go through
    these two helpers so that a launch tape being recorded (radet_amd/tape.py) sees them."""
    if stream is None:
        ev.record()
    else:
        ev.record(stream)
    t = _lib.TAPE
    if t is not None:
        t.record_event(ev, _stream() if stream is None else stream.cuda_stream)


def ev_wait(ev, stream=None):
    """`stream` (None: the current one) waits for event ev"""
    (torch.cuda.current_stream() if stream is None else stream).wait_event(ev)
    t = _lib.TAPE
    if t is not None:
        t.wait_event(ev, _stream() if stream is None else stream.cuda_stream)


def fill_zero(t):
    """stream-ordered zero fill of a contiguous device tensor (hipMemsetAsync; tape-able, unlike Tensor.zero_())"""
    _lib.call("radet_fill_zero", _ptr(t), C.c_size_t(t.numel() * t.element_size()), _stream())


def copy_d2d(dst, src):
    """stream-ordered device-to-device copy between contiguous tensors of equal byte size"""
    n = dst.numel() * dst.element_size()
    assert n == src.numel() * src.element_size()
    _lib.call("radet_copy_d2d", _ptr(dst), _ptr(src), C.c_size_t(n), _stream())


class Levels:
    """Row-concatenated multi-level NHWC geometry: level l = B images of h_l x w_l pixels."""

    def __init__(self, hw, B):
        self.hw = [(int(h), int(w)) for h, w in hw]
        self.B = int(B)
        self.offsets = []
        r = 0
        for h, w in self.hw:
            self.offsets.append(r)
            r += self.B * h * w
        self.rows = r

    def __len__(self):
        return len(self.hw)

    def conv_out(self, k, stride, pad):
        return Levels([((h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1) for h, w in self.hw], self.B)

    def level_rows(self, l):
        h, w = self.hw[l]
        return self.offsets[l], self.offsets[l] + self.B * h * w

    def sub(self, l):
        return Levels([self.hw[l]], self.B)


def _desc(pairs):
    """pairs: list of (Hi, Wi, Ho, Wo, in_off, out_off) -> ctypes int array"""
    flat = [int(v) for p in pairs for v in p]
    return (C.c_int * len(flat))(*flat), len(pairs)


class _LRU(dict):
    """dict with a size bound: the oldest insertions are dropped first (a dropped gather table / tune entry is simply
    rebuilt on its next use)."""

    def __init__(self, cap):
        super().__init__()
        self.cap = cap

    def __setitem__(self, k, v):
        super().__setitem__(k, v)
        while len(self) > self.cap:
            del self[next(iter(self))]


_TABLE_CACHE = _LRU(1024)      # gather tables / parity classes per conv geometry (~80 per (B, H, W) plan)

# Measurement hook (bench.py): when EVENTS is a list, every conv GEMM launch issued through this module is bracketed by a
# pair of HIP events on the stream it is launched on and appended as dict(key, flops, start, end, replay).  `key` names
# the kernel instantiation the launcher dispatches to, `flops` = 2 * rows * Cout * Cin * taps of the launch (x 2 for a
# grouped pair), `replay()` issues the same launch again (same buffers) for the "alone on the device" timing.
EVENTS = None
STAGE = ""          # part of the detector the engine is enqueueing (stem, layer1..4, neck, head): recorded with every event
_TILES = {1: "128, 128, 2, 2", 2: "128, 64, 2, 2", 3: "64, 64, 2, 2", 4: "128, 32, 4, 1", 5: "128, 128, 2, 4", 6: "256, 128, 4, 2",
          7: "64, 64, 2, 2", 8: "64, 64, 2, 2"}


def _igemm_key(t, x):
    """kernel symbol an implicit-GEMM launch with tile_override word t dispatches to (conv_igemm.hip: igemm_impl)"""
    tid = t & 0xFF
    if tid == 0:
        return "conv_igemmg_kernel<launcher heuristic>"
    if t & H2 and (t & P3 or t & X3):
        if t & P3 and tid == 7:                 # K-divided 64 x 64 tile on plane pairs (round 6)
            return "conv_igemmg_kernel<64, 64, 2, 2, 232, 64, 2, false>"
        if t & P3 and t & ROWPAIRS and tid in (5, 6):
            return f"conv_igemmg_kernel<{_TILES[tid]}, {208 | ((t >> 8) & 1)}, 32, 2, false>"
        if t & P3:
            tag, bk = 80 | ((t >> 8) & 1), 16
        else:
            tag, bk = 72 | ((t >> 8) & 1), 32
            if tid >= 7:
                nst = 2 if tid == 7 else (4 if t & STAGES4 else (3 if t & STAGES3 else 2))
                return f"conv_igemmg_kernel<{_TILES[tid]}, {(tag | 32) & ~1}, {64 if tid == 7 else 32}, {nst}, false>"
        return f"conv_igemmg_kernel<{_TILES.get(tid, '?')}, {tag}, {bk}, {3 if t & STAGES3 else 2}, false>"
    if t & P3:
        tag, bk = 16 | ((t >> 8) & 1), 16
    elif t & STORE_BF16:
        tag, bk = 4 | ((t >> 8) & 1), 32 if t & 0x200 else 16
    elif t & X3:
        tag, bk = 8 | ((t >> 8) & 1), 32
        if tid >= 7:                        # K-divided 64 x 64 tiles: 7 = four k-groups of a 64-channel stage,
            tag, bk = tag | 32, 64 if tid == 7 else 32          # 8 = two k-groups x two column halves of 32 channels
            return f"conv_igemmg_kernel<{_TILES[tid]}, {tag}, {bk}, 2, false>"
    else:
        tag, bk = ((t >> 8) & 1) | (2 if t & MATH_BF16 else 0), 32 if t & 0x200 else 16
    skw = (t >> 20) & 7
    stages = 3 if (t & STAGES3) and (tag < 2 or tag & 24) and not skw else 2
    return f"conv_igemmg_kernel<{_TILES.get(tid, '?')}, {tag}, {bk}, {stages}, {'true' if skw and tag == 0 else 'false'}>"


def _timed(key, flops, fn, nbytes=0.0, kind="fwd", geom=None):
    """key: the kernel's name, or a callable producing it (only evaluated when a measurement is running); kind: fwd / dgrad /
    wgrad; the engine's STAGE at the time of the call goes into the record (bench.py's per-stage table)"""
    ev = EVENTS
    if ev is None:
        return fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    fn()
    e.record()
    ev.append(dict(key=key() if callable(key) else key, flops=float(flops), bytes=float(nbytes), start=s, end=e, replay=fn,
                   stage=STAGE, kind=kind, geom=geom))


def _conv_bytes(g, groups=1):
    """algorithmic HBM bytes of one conv launch at fp32: input once + weights + output once"""
    return 4.0 * groups * (g.lin.rows * g.cin + g.cout * g.k * g.k * g.cin + g.lout.rows * g.cout)

STRIDED_DGRAD_CLASSES = True   # parity-class dgrad for strided convs (False = one dense launch)
IDENTITY_NO_TABLE = os.environ.get("RADET_IDENTITY_TABLE", "0") != "1"   # 1 x 1 / stride 1 convs run without a gather table


def _gather_table(key, B, k, so, sr, off, div, desc, nseg, rows):
    """Device gather table for one conv geometry, shared by every conv with the same geometry."""
    dev = torch.cuda.current_device()
    key = (dev,) + key
    t = _TABLE_CACHE.get(key)
    if t is None:
        mp = _lib.load().radet_gather_table_rows(rows)
        t = torch.empty(k * k * mp, dtype=torch.int32, device=torch.device("cuda", dev))
        _lib.call("radet_build_gather_table", _ptr(t), B, k, k, so, sr, off, div, desc, nseg, _stream())
        # Tables are shared by every conv of this geometry, whatever stream it launches on (the cls tower's dgrad builds
        # the table on the main stream, the reg tower's dgrad picks it out of the cache and launches on the side
        # stream a few microseconds later): finish the build before anyone can see the table.  Once per geometry.
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream().synchronize()
        _TABLE_CACHE[key] = t
    return t


class ConvGeom:
    """Geometry of one convolution over a multi-level input (same weights on every level)."""

    def __init__(self, lin, cin, cout, k, stride, pad):
        self.lin, self.cin, self.cout, self.k, self.stride, self.pad = lin, cin, cout, k, stride, pad
        self.lout = lin.conv_out(k, stride, pad)
        self.B = lin.B
        # the kernels' tile loads address a tensor with 32-bit byte offsets (buffer_load ... lds, round 6): 4 GiB - 256 B per tensor
        # (5.7x the largest tensor of the reference config at samples_per_gpu = 16: 314 MB)
        big = max(lin.rows * cin, self.lout.rows * cout) * 4
        if big >= (1 << 32) - 256:
            raise _lib.RadetHipError(f"conv geometry {cin}->{cout} k{k} over {lin.rows} rows: a {big / 2 ** 30:.1f} GiB activation tensor "
                                     "exceeds the 4 GiB the MI355X kernels address per tensor; use a smaller batch per GPU")
        f, b = [], []
        for (hi, wi), (ho, wo), io, oo in zip(lin.hw, self.lout.hw, lin.offsets, self.lout.offsets):
            f.append((hi, wi, ho, wo, io, oo))
            b.append((ho, wo, hi, wi, oo, io))   # dgrad: rows over the conv INPUT grid, gather from dy
        self.fwd_desc, self.nseg = _desc(f)
        self.bwd_desc, _ = _desc(b)
        self._key = (tuple(lin.hw), tuple(lin.offsets), lin.B, k, stride, pad)
        self.fwd_tile = self.bwd_tile = 0     # 0 = launcher heuristic; set by autotune()
        self.math = 0                         # 1: bf16 math mode (operands rounded to bf16, fp32 accumulate)
        self.wgrad_flags = 0                  # tile override of the wgrad launcher (set by autotune_wgrad)
        self.h16 = False                      # bf16 tensors in HBM (tuning runs use the matching kernels)
        self._x3 = False                      # fp32 tensors, conv products from 16-bit planes of the operands (tile flag X3) ...
        self.h2 = False                       # ... two fp16 planes and 3 plane products (tile flag H2) instead of three bf16 / 6
        self.nsplit = _lib.load().radet_conv2d_wgrad_splits(self.lout.rows, cin, cout, k, k)
        self.pairs = False                    # forward launches read x as plane pairs written by its producer (fwd_tile_q)
        self.fwd_tile_q = 0
        self.bwd_tile_q = 0                   # dgrad launches whose dy arrives as plane pairs (pairs-only gradients, round 6)
        self.wgrad_pair_flags = 0             # weight gradient on fp16 plane pairs: 0x40 one-tap kernel, bits 4-5 its tile (1: 128 x 128)
        self.nsplit_pairs = 0                 # its pixel splits (0: nsplit)
        self._ft = self._bt = self._classes = None
        self._cgroup = 0                                   # 0 = not looked at yet, None = no class launch for this geometry

    @property
    def x3(self):
        return self._x3

    @x3.setter
    def x3(self, v):                          # True: the bf16-plane arithmetic; "h2": the fp16 hi / lo arithmetic
        self._x3 = bool(v)
        if v == "h2":
            self.h2 = True

    # The geometry keeps its tables alive: _TABLE_CACHE only dedupes them across geometries and may drop its own
    # reference at any time -- a table freed while a launch on a side stream still reads it would be recycled by the
    # caching allocator and overwritten under that launch.
    @property
    def identity(self):
        """1 x 1 / stride 1 / no padding: GEMM row m reads row m (input and output levels coincide) -- no gather table, the
        launchers take NULL and save the dependent table load in every workgroup's prologue"""
        return self.k == 1 and self.stride == 1 and self.pad == 0 and IDENTITY_NO_TABLE

    @property
    def fwd_table(self):
        if self.identity:
            return None
        if self._ft is None:
            self._ft = _gather_table(("f",) + self._key, self.B, self.k, self.stride, 1, -self.pad, 1, self.fwd_desc,
                                     self.nseg, self.lout.rows)
        return self._ft

    @property
    def bwd_table(self):
        if self.identity:
            return None
        if self._bt is None:
            self._bt = _gather_table(("b",) + self._key, self.B, self.k, 1, -1, self.pad, self.stride, self.bwd_desc,
                                     self.nseg, self.lin.rows)
        return self._bt


_SPLITK_WS = {}


def splitk_ws_for(stream):
    """Create the split-K workspace of `stream` now (e.g. before a hipGraph capture on it)."""
    with torch.cuda.stream(stream):
        return splitk_ws()


def splitk_ws():
    """One split-K workspace per (device, stream): launches on one stream are ordered, so all convs issued
    there can share it; concurrent streams get their own."""
    dev = torch.cuda.current_device()
    key = (dev, torch.cuda.current_stream().cuda_stream)
    t = _SPLITK_WS.get(key)
    if t is None:
        # 64 MiB; zero-initialised: the first words are the arrival tickets of the in-launch split-K reduction
        t = torch.zeros(16 * 1024 * 1024, device=torch.device("cuda", dev))
        _SPLITK_WS[key] = t
    return t


def _strided_dgrad_classes(g):
    """Parity classes of the dgrad of a strided conv (built once per geometry).

    dx[h, w] only receives taps r with (h + pad - r) % stride == 0, so the rows of the dgrad GEMM split
    into stride^2 classes by (h % stride, w % stride), each with a fixed subset of contributing taps.
    One tap-subset launch per class does only the non-zero work (a dense launch wastes 75 % of the MFMAs
    of a 3x3/2 conv).  Returns a list of dicts(table, out_rows, tap_ids, ntaps, rows, zero)."""
    import numpy as np
    if g._classes is not None:
        return g._classes
    key = ("cls",) + g._key
    dev = torch.cuda.current_device()
    ck = (dev,) + key
    if ck in _TABLE_CACHE:
        g._classes = _TABLE_CACHE[ck]
        return g._classes
    KT = g.k * g.k
    M = g.lin.rows
    mp = _lib.load().radet_gather_table_rows(M)
    full = g.bwd_table.cpu().numpy().reshape(KT, mp)[:, :M]
    cls_id = np.empty(M, np.int64)
    for (h, w), off in zip(g.lin.hw, g.lin.offsets):
        hh, ww = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        c = ((hh % g.stride) * g.stride + (ww % g.stride)).reshape(-1)
        cls_id[off:off + g.B * h * w] = np.tile(c, g.B)
    out, zero_rows = [], []
    for c in range(g.stride * g.stride):
        rows = np.nonzero(cls_id == c)[0]
        if rows.size == 0:
            continue
        taps = [t for t in range(KT) if (full[t, rows] >= 0).any()]
        if not taps:
            zero_rows.append(rows)
            continue
        mpc = _lib.load().radet_gather_table_rows(rows.size)
        tab = np.full((len(taps), mpc), -1, np.int32)
        tab[:, :rows.size] = full[taps][:, rows]
        out.append(dict(table=torch.from_numpy(tab).cuda(), out_rows=torch.from_numpy(rows.astype(np.int32)).cuda(),
                        tap_ids=(C.c_int * len(taps))(*taps), ntaps=len(taps), rows=int(rows.size), zero=False))
    if zero_rows:
        rows = np.concatenate(zero_rows)
        mpc = _lib.load().radet_gather_table_rows(rows.size)
        tab = np.full((1, mpc), -1, np.int32)
        out.append(dict(table=torch.from_numpy(tab).cuda(), out_rows=torch.from_numpy(rows.astype(np.int32)).cuda(),
                        tap_ids=(C.c_int * 1)(0), ntaps=1, rows=int(rows.size), zero=True))
    if not torch.cuda.is_current_stream_capturing():
        torch.cuda.current_stream().synchronize()      # the uploads above have landed before another stream can use them
    _TABLE_CACHE[ck] = g._classes = out
    return out


STRIDED_DGRAD_GROUP = os.environ.get("RADET_DGRAD_CLASS_GROUP", "1") != "0"


def _strided_dgrad_group(g):
    """All tap-carrying parity classes of a strided dgrad as ONE launch (radet_conv2d_igemm_classes): rows sorted by
    class (most taps first), each class padded to 128 rows.  None when the geometry has a single class (1x1 / 2) or a
    class with more than 4 taps.  Returns dict(table [kmax][M], out_rows [M], tap_ids[16], ntaps[4], start[4], ncls, M)."""
    import numpy as np
    if g._cgroup is None or g._cgroup != 0:
        return g._cgroup
    cls = sorted((c for c in _strided_dgrad_classes(g) if not c["zero"]), key=lambda c: -c["ntaps"])
    grp = None
    if STRIDED_DGRAD_GROUP and 2 <= len(cls) <= 4 and all(c["ntaps"] <= 4 for c in cls):
        ck = (torch.cuda.current_device(), "clsgrp") + g._key
        grp = _TABLE_CACHE.get(ck)
        if grp is None:
            kmax = max(c["ntaps"] for c in cls)
            starts, m = [], 0
            for c in cls:
                starts.append(m)
                m += -(-c["rows"] // 128) * 128
            tab = np.full((kmax, _lib.load().radet_gather_table_rows(m)), -1, np.int32)   # row stride = the launcher's Mp
            orow = np.full(m, -1, np.int32)
            tids = [0] * 16
            for i, (c, s0) in enumerate(zip(cls, starts)):
                t = c["table"].cpu().numpy()
                tab[:c["ntaps"], s0:s0 + c["rows"]] = t[:, :c["rows"]]
                orow[s0:s0 + c["rows"]] = c["out_rows"].cpu().numpy()
                tids[4 * i:4 * i + c["ntaps"]] = list(c["tap_ids"])
            pad4 = lambda v: list(v) + [0] * (4 - len(v))  # noqa: E731
            grp = dict(table=torch.from_numpy(tab).cuda(), out_rows=torch.from_numpy(orow).cuda(),
                       tap_ids=(C.c_int * 16)(*tids), ntaps=(C.c_int * 4)(*pad4([c["ntaps"] for c in cls])),
                       start=(C.c_int * 4)(*pad4(starts)), ncls=len(cls), M=m)
            if not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream().synchronize()
            _TABLE_CACHE[ck] = grp
    g._cgroup = grp
    return grp


_TUNE_CACHE = _LRU(8192)
_TUNE_DIRTY = False
TUNE_RUNS = 0                  # number of shapes actually timed in this process (tests: no re-tune after the first pass)


# Tile / split choices are persisted per conv geometry:
#   1. radet_amd/tune_gfx950.json -- shipped with the package: the choices for the standard geometries (r50 640x480 at
#      B = 1, 2, 4, 8 in the three arithmetic modes, r101 800x800 B = 2), produced by tools/make_tune.py on an MI355X, so
#      that identical runs pick identical tiles (reproducible losses, no start-up tuning);
#   2. the user file ($RADET_TUNE_FILE, default ~/.cache/radet_amd/tune_gfx950.json) -- shapes tuned on this machine.
# Unknown shapes are timed once and appended to the user file.  RADET_AUTOTUNE=0 disables tuning AND the files
# (launcher heuristics only).
_PACKAGED_TUNE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tune_gfx950.json")
_TUNE_LOADED = False


def _tune_file():
    return os.environ.get("RADET_TUNE_FILE") or os.path.join(os.path.expanduser("~"), ".cache", "radet_amd", "tune_gfx950.json")


def _read_tune(path):
    import ast
    import json
    with open(path) as fh:
        d = json.load(fh)
    return ({ast.literal_eval(k): tuple(v) for k, v in d.get("igemm", {}).items()},
            {ast.literal_eval(k): tuple(v) for k, v in d.get("wgrad", {}).items()})


def load_tune_cache():
    global _TUNE_LOADED
    if _TUNE_LOADED:
        return
    _TUNE_LOADED = True
    for path in (_PACKAGED_TUNE, _tune_file()):
        if os.path.exists(path):
            try:
                a, b = _read_tune(path)
            except (ValueError, SyntaxError, OSError):
                continue                      # unreadable cache: tune again
            for k, v in a.items():
                _TUNE_CACHE[k] = v
            for k, v in b.items():
                _WTUNE_CACHE[k] = v


def save_tune_cache(path=None):
    """Write every known choice (packaged + tuned here) to the user file; silent if the location is not writable."""
    import json
    global _TUNE_DIRTY
    if not _TUNE_DIRTY and path is None:
        return
    f = path or _tune_file()
    try:
        os.makedirs(os.path.dirname(f) or ".", exist_ok=True)
        tmp = f + f".{os.getpid()}.tmp"
        with open(tmp, "w") as fh:
            json.dump(dict(igemm={repr(k): list(v) for k, v in _TUNE_CACHE.items()},
                           wgrad={repr(k): list(v) for k, v in _WTUNE_CACHE.items()}), fh, indent=0, sort_keys=True)
        os.replace(tmp, f)
        _TUNE_DIRTY = False
    except OSError:
        pass


TUNE_REPS = int(os.environ.get("RADET_TUNE_REPS", "3"))      # timed samples per candidate (the fastest counts)
# launches per sample, back to back between one pair of events: a single 30-60 us launch per sample carries 5-10 us of
# event / launch-gap jitter, more than the differences between the candidates
TUNE_BURST = max(1, int(os.environ.get("RADET_TUNE_BURST", "4")))


def autotune(g, need_dgrad=True, reps=None):
    """Pick the fastest (block tile, K step) of the implicit-GEMM kernel for this geometry by timing the
    candidates once (results cached per shape, so identical layers and identical models agree)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    reps = reps or TUNE_REPS

    def best_of(fn, cands):
        """fastest candidate: `reps` interleaved rounds over all candidates (clock / cache drift hits every candidate
        alike), each candidate scored by its fastest round"""
        for t in cands:
            fn(t)
        torch.cuda.synchronize()
        best = [float("inf")] * len(cands)
        for _ in range(reps):
            for i, t in enumerate(cands):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _b in range(TUNE_BURST):
                    fn(t)
                e.record()
                e.synchronize()
                best[i] = min(best[i], s.elapsed_time(e) / TUNE_BURST)
        return cands[min(range(len(cands)), key=lambda i: (best[i], i))]

    def cands(kdim, n, m, taps):
        tiles = [4] if n <= 32 else [1, 2, 3]
        c = [t for t in tiles]
        if kdim % 32 == 0:
            c += [t | 0x200 for t in tiles]
        # short grids: also try explicit split-K factors (bits 12-15) instead of the launcher's heuristic
        if getattr(g, "x3", False) and not g.math and not g.h16 and n > 32:
            # K-divided 64 x 64 tiles (the waves share the operand splits): K steps of 64 / 32 channels
            c += ([7] if kdim % 64 == 0 else []) + ([8] if kdim % 32 == 0 else [])
            if _h2key(g) and kdim % 32 == 0 and KW_DEEP:
                # fp16 hi / lo arithmetic: tile 8 with loads two / three stages ahead (16 KiB of LDS per stage)
                c += [8 | STAGES3, 8 | STAGES4]
        out = list(c)
        for t in c:
            bm = 64 if (t & 0xFF) in (3, 7, 8) else 128
            bn = {1: 128, 2: 64, 3: 64, 4: 32, 7: 64, 8: 64}[t & 0xFF]
            ntiles = -(-m // bm) * -(-n // bn)
            nk = taps * kdim // (64 if (t & 0xFF) == 7 else (32 if (t & 0x200 or (t & 0xFF) == 8) else 16))
            if ntiles < 1024:
                out += [t | (sk << 12) for sk in (1, 2, 3, 4, 5, 6, 8) if nk // sk >= 4]
                # stream-K (fp32 tensors, fp32 / bf16-rounded math is decided by the launcher's tag: fp32 only)
                if not g.math and not g.h16 and (t & 0xFF) in (2, 3, 4) and ntiles % 256:      # (not the K-divided tiles)
                    out += [t | (w * STREAMK) for w in (1, 2, 3, 4) if ntiles * nk >= 256 * w]
        return out

    global _TUNE_DIRTY, TUNE_RUNS
    key = (g._key, g.cin, g.cout, g.math, g.h16) + (("x3",) if getattr(g, "x3", False) else ()) + _h2key(g)
    if key not in _TUNE_CACHE:
        _TUNE_DIRTY = True
        TUNE_RUNS += 1
        dt = torch.bfloat16 if g.h16 else torch.float32
        x = torch.randn(g.lin.rows, g.cin, device=dev).to(dt)
        w = (torch.randn(g.cout * g.k * g.k * g.cin, device=dev) * 0.05).to(dt)
        y = torch.empty(g.lout.rows, g.cout, device=dev, dtype=dt)
        tmp_keys = _tune_slots(g, x, w, outputs=(y,))
        fc = cands(g.cin, g.cout, g.lout.rows, g.k * g.k)
        if not g.math and not g.h16:      # forward launches run alone on the device: 3 LDS stages may pay (0x20000)
            fc = fc + [t | STAGES3 for t in fc if (t & 0xFF) < 7 and not t & (STAGES3 | STAGES4)]
        ft = best_of(lambda t: conv_fwd(g, x, w, None, y, relu=True, tile=t), fc)
        bt = 0
        if need_dgrad and g.cout % 16 == 0:
            dx = torch.empty(g.lin.rows, g.cin, device=dev, dtype=dt)
            tmp_keys += _tune_slots(g, outputs=(dx,))
            # (no 3-stage candidates for dgrad: timed alone they win, next to the wgrad streams they lose -- a tune file
            # that allowed them made the step 1 % slower)
            bt = best_of(lambda t: conv_dgrad(g, y, w, dx, mask=x, tile=t), cands(g.cout, g.cin, g.lin.rows, g.k * g.k))
        _TUNE_CACHE[key] = (ft, bt)
        unregister_amax(tmp_keys)
        if os.environ.get("RADET_TUNE_LOG"):
            print(f"[tune igemm] M={g.lout.rows} {g.cin}->{g.cout} k{g.k}s{g.stride}: fwd tile={ft:#x} dgrad tile={bt:#x}")
    g.fwd_tile, g.bwd_tile = _TUNE_CACHE[key]
    if getattr(g, "pairs", False) and _h2key(g) and g.cin % 32 == 0:
        # forward launches whose x arrives as fp16 plane pairs (the producer's epilogue wrote them): the 4-wave plane tiles
        keyq = key + ("q",)
        if keyq not in _TUNE_CACHE:
            _TUNE_DIRTY = True
            TUNE_RUNS += 1
            x = torch.randn(g.lin.rows, g.cin, device=dev)
            w = torch.randn(g.cout * g.k * g.k, g.cin, device=dev) * 0.05
            xq, wq = Planes.from_float(x, kind="h2"), Planes.from_float(w, kind="h2")
            y = torch.empty(g.lout.rows, g.cout, device=dev)
            cq = [1, 2, 3, 1 | STAGES3, 2 | STAGES3, 3 | STAGES3] + ([7] if g.cin % 64 == 0 else [])
            for t in list(cq):
                bm = 64 if (t & 0xFF) in (3, 7) else 128
                bn = {1: 128, 2: 64, 3: 64, 7: 64}[t & 0xFF]
                ntiles = -(-g.lout.rows // bm) * -(-g.cout // bn)
                nk = g.k * g.k * g.cin // (64 if (t & 0xFF) == 7 else 32)
                if ntiles < 1024:
                    cq += [t | (sk << 12) for sk in (1, 2, 3, 4, 5, 6, 8) if nk // sk >= 4]
            ftq = best_of(lambda t: conv_fwd(g, xq, wq, None, y, relu=True, tile=t), cq)
            btq = 0
            if need_dgrad and g.stride == 1 and g.cout % 32 == 0:
                # ... and dgrad launches whose dy arrives as pairs (with the ReLU mask read from a pair tensor)
                dyq = Planes.from_float(torch.randn(g.lout.rows, g.cout, device=dev), kind="h2")
                wtq = Planes.from_float(torch.randn(g.cin * g.k * g.k, g.cout, device=dev) * 0.05, kind="h2")
                dx = torch.empty(g.lin.rows, g.cin, device=dev)
                tmp_q = _tune_slots(g, outputs=(dx,))
                cb = [1, 2, 3] + ([7] if g.cout % 64 == 0 else [])
                for t in list(cb):
                    bm = 64 if (t & 0xFF) in (3, 7) else 128
                    bn = {1: 128, 2: 64, 3: 64, 7: 64}[t & 0xFF]
                    ntiles = -(-g.lin.rows // bm) * -(-g.cin // bn)
                    nk = g.k * g.k * g.cout // (64 if (t & 0xFF) == 7 else 32)
                    if ntiles < 1024:
                        cb += [t | (sk << 12) for sk in (1, 2, 3, 4, 5, 6, 8) if nk // sk >= 4]
                btq = best_of(lambda t: conv_dgrad(g, dyq, wtq, dx, mask=xq, tile=t), cb)
                unregister_amax(tmp_q)
            _TUNE_CACHE[keyq] = (ftq, btq)
            if os.environ.get("RADET_TUNE_LOG"):
                print(f"[tune igemm pairs] M={g.lout.rows} {g.cin}->{g.cout} k{g.k}s{g.stride}: fwd tile={ftq:#x} dgrad tile={btq:#x}")
        g.fwd_tile_q, g.bwd_tile_q = _TUNE_CACHE[keyq]


KW_DEEP = os.environ.get("RADET_KW_DEEP", "1") != "0"


def _h2key(g):
    return ("h2",) if (getattr(g, "h2", False) and getattr(g, "x3", False) and not g.math and not g.h16) else ()


def _tune_slots(g, *tensors, outputs=()):
    """amax slots for the tuner's scratch tensors (fp16 hi / lo arithmetic): the operands' slots are MEASURED once (a constant
    would overflow fp16 for a K or an init whose sums leave its range), the outputs' start at zero and are raised by the timed
    launches' epilogues; registered by address, so that the timed launches do not each run a stand-alone absmax pass.
    Returns the registry keys."""
    keys = []
    if _h2key(g):
        for t in tensors:
            s = new_amax(t.device)
            if t.is_floating_point() and t.dtype == torch.float32 and t.is_contiguous() and t.numel() % 4 == 0:
                absmax(t, s)
            elif t.is_floating_point() and t.dtype == torch.float32:
                s.fill_(int(t.abs().max().view(torch.int32)) if t.numel() else 0)
            keys.append(register_amax(t, s))
        for t in outputs:
            t.zero_()                               # (an uninitialised output may hold NaN bit patterns: they must not be read back)
            keys.append(register_amax(t, new_amax(t.device)))
    return keys


MATH_BF16 = 0x400      # tile_override bit of the implicit-GEMM entry points


STAGES3 = 0x20000                         # 3 LDS stages in the fp32 implicit-GEMM kernel (forward launches)
STAGES4 = 0x40000                         # 4 LDS stages (fp16 hi / lo arithmetic, K-divided tile 8)
X3 = 0x1000000                            # fp32 tensors, products from three bf16 planes per operand on the bf16 matrix cores
STREAMK = 0x100000                        # * w (1..7): stream-K schedule with w persistent workgroups per CU (fp32 tags)
STORE_BF16, OUT_F32 = 0x800, 0x10000      # bf16 tensors in HBM / fp32 output from bf16 inputs (predictor heads)


P3 = 0x2000000                            # x / w arrive as plane tensors (bf16 triples, or fp16 pairs with H2); fp32 outputs
P3_BK8 = 0x4000000                        # ... with a K step of 16 instead of 32 channels
H2 = 0x8000000                            # fp16 hi / lo arithmetic (with X3 or P3): 3 f16 MFMAs per K = 16 step, needs amax slots
ROWPAIRS = 0x80000                        # plane-pair operands, 8-wave tiles: both planes of a tile row in one 128-byte piece per load
MASKQ = 0x10000000                        # the launch's ReLU mask tensor is an fp16 plane-pair tensor (round 6: pairs-only activations)


# ---------------------------------------------------------------------- amax slots (fp16 hi / lo arithmetic, radet_hip.h)
# A slot is AMAX_WORDS (64) int32 words of device memory; its value -- the LARGEST word, read as the bit pattern of a float -- is
# >= every |element| of "its" tensor (64 words: thousands of waves raise a slot at once, include/radet_hip.h).  The
# engine registers the slots of its buffers here (activations by storage, so that row slices of a buffer resolve to the
# buffer's slot; weights by exact address, they are views of one arena); the conv launchers look the operands' slots up.
# A tensor without a slot (ad-hoc calls: tests, the autotuner) gets one computed on the spot by a stand-alone pass.
AMAX_WORDS = 64
_AMAX_EXACT = {}
_AMAX_STORAGE = {}


def new_amax(device, n=None):
    """a zeroed slot (int32 [radet_amax_slot_words()]), or n of them ([n, words])"""
    global AMAX_WORDS
    AMAX_WORDS = _lib.load().radet_amax_slot_words()
    return torch.zeros((AMAX_WORDS,) if n is None else (n, AMAX_WORDS), dtype=torch.int32, device=device)


def amax_value(slot):
    """the float a slot stands for (host synchronisation: tests / debugging)"""
    return float(slot.max().view(torch.float32))


def register_amax(t, slot, by_storage=False):
    """slot: a new_amax() tensor.  Returns the registry key (for unregister_amax).  The entry is only honoured while `t` is
    alive: the registry is keyed by device address, and the caching allocator hands a dead tensor's address to the next one."""
    import weakref
    if by_storage:
        k = ("s", t.untyped_storage().data_ptr())
        _AMAX_STORAGE[k[1]] = (slot, weakref.ref(t))
    else:
        k = ("e", t.data_ptr())
        _AMAX_EXACT[k[1]] = (slot, weakref.ref(t))
    return k


def _live(table, ptr):
    ent = table.get(ptr)
    if ent is None:
        return None
    if ent[1]() is None:                    # the registered tensor is gone: whoever lives at this address now is someone else
        del table[ptr]
        return None
    return ent


def unregister_amax(keys):
    for kind, ptr in keys:
        (_AMAX_STORAGE if kind == "s" else _AMAX_EXACT).pop(ptr, None)
    _AMAX_MEMO.clear()
    _SCALES.clear()


_AMAX_MEMO = {}


def amax_slot(t, compute=False):
    """the registered slot of tensor / Planes t (or of the buffer it is a slice of); None, or -- with compute=True -- a
    fresh slot filled by radet_absmax"""
    if t is None:
        return None
    if _isp(t):
        return t.amax
    p = t.data_ptr()
    ent = _live(_AMAX_EXACT, p)
    if ent is None:
        ent = _live(_AMAX_MEMO, p)
        if ent is None and _AMAX_STORAGE:
            ent = _live(_AMAX_STORAGE, t.untyped_storage().data_ptr())
            if ent is not None:
                _AMAX_MEMO[p] = ent         # (a row slice of a registered buffer: remembered by its own address)
    s = ent[0] if ent is not None else None
    if s is None and compute:
        s = new_amax(t.device)
        absmax(t, s)
    return s


def absmax(t, slot):
    """raise `slot` to the largest magnitude of the fp32 tensor t (contiguous, numel % 4 == 0)"""
    assert t.dtype == torch.float32 and t.is_contiguous()
    _lib.call("radet_absmax", _ptr(t), C.c_size_t(t.numel()), _ptr(slot), _stream())


_SCALES = _LRU(4096)


def _scales(x, w, y, x1=None, w1=None, y1=None, need=True, yq=None, wmeta=None, addend=None):
    """RadetScales for a launch (cached per slot combination).  need: the x / w slots are required (h2 arithmetic): missing
    ones are computed on the spot; otherwise only the output slot matters and None is returned when there is none.
    yq (Planes "h2"): the pair copy of the output, with wmeta = (w_l1 slot, bias amax slot or None) of the conv and the
    launch's addend (its slot bounds the residual term)."""
    sx, sw, sy = amax_slot(x, need), amax_slot(w, need), amax_slot(y)
    if y is None and yq is not None:       # pairs-only output: the slot of its TRUE largest magnitude is still raised
        sy = getattr(yq, "true_amax", None)
    sx1, sw1, sy1 = amax_slot(x1, need), amax_slot(w1, need), amax_slot(y1)
    if not need and sy is None and sy1 is None and yq is None:
        return None
    ext = (None,) * 6
    if yq is not None:
        xt = getattr(x, "true_amax", None) if _isp(x) else amax_slot(x, True)
        assert xt is not None and wmeta is not None, "pair copy of a conv output: the true amax slot of x and the conv's L1 slot"
        ext = (yq.t, yq.amax, xt, wmeta[0], wmeta[1], amax_slot(addend, True) if addend is not None else None)
    slots = (sx, sw, sy, sx1, sw1, sy1) + ext
    key = tuple(0 if t is None else t.data_ptr() for t in slots)
    ent = _SCALES.get(key)
    if ent is None:
        sc = _lib.RadetScales(*[None if v == 0 else v for v in key])
        ent = _SCALES[key] = (C.byref(sc), sc, slots)              # (the struct and the slots stay alive with it)
    return ent[0]


class Planes:
    """A [rows, C] fp32 tensor in the operand format of the conv GEMMs (include/radet_hip.h).  kind "b3": bf16 plane triples,
    `t` is a bf16 tensor [rows, 3 * C] whose row r holds hi | mid | lo with hi + mid + lo == the fp32 value exactly.
    kind "h2": fp16 plane pairs, `t` is an fp16 tensor [rows, 2 * C] (32-channel groups [hi | lo]) of the values scaled by
    the power of two of `amax` (the tensor's amax slot, see new_amax).  Only the conv GEMMs read it."""

    def __init__(self, rows, C, device=None, t=None, kind="b3", amax=None):
        self.rows, self.C, self.kind = int(rows), int(C), kind
        if t is None:
            t = (torch.empty(self.rows, 2 * self.C, device=device, dtype=torch.float16) if kind == "h2"
                 else torch.empty(self.rows, 3 * self.C, device=device, dtype=torch.bfloat16))
        self.t = t
        self.amax = amax
        if kind == "h2" and amax is None:
            self.amax = new_amax(t.device)

    def __getitem__(self, sl):
        assert isinstance(sl, slice) and sl.step in (None, 1)
        v = self.t[sl]
        return Planes(v.shape[0], self.C, t=v, kind=self.kind, amax=self.amax)

    def data_ptr(self):
        return self.t.data_ptr()

    def to_float(self):
        out = torch.empty(self.rows, self.C, device=self.t.device)
        merge_planes(self, out)
        return out

    @staticmethod
    def from_float(x, kind="b3"):
        p = Planes(x.shape[0], x.shape[1], device=x.device, kind=kind)
        split_planes(x, p)
        return p


def _isp(t):
    return isinstance(t, Planes)


def _ptr_any(t):
    return _ptr(t.t) if _isp(t) else _ptr(t)


def split_planes(src, dst, src_amax=None):
    """dst (Planes) = plane split of the 2-D fp32 tensor src (row stride may exceed the width): exact bf16 triples, or fp16
    pairs scaled by the power of two of src's amax slot (src_amax, else the registered one, else computed here)"""
    assert src.dim() == 2 and src.dtype == torch.float32 and src.stride(1) == 1 and dst.C == src.shape[1]
    if dst.kind == "h2":
        sa = src_amax if src_amax is not None else amax_slot(src)
        if sa is None:
            sa = new_amax(src.device)
            absmax(src.contiguous(), sa)
        _lib.call("radet_split_pairs", src.data_ptr(), _ptr(dst.t), C.c_size_t(src.shape[0]), src.shape[1], src.stride(0),
                  _ptr(sa), _ptr(dst.amax), _stream())
        return
    _lib.call("radet_split_planes", C.c_void_p(src.data_ptr()), _ptr(dst.t), C.c_size_t(src.shape[0]), src.shape[1],
              src.stride(0), _stream())


def merge_planes(src, dst):
    assert dst.dim() == 2 and dst.dtype == torch.float32 and dst.stride(1) == 1 and src.C == dst.shape[1]
    if src.kind == "h2":
        _lib.call("radet_merge_pairs", _ptr(src.t), dst.data_ptr(), C.c_size_t(dst.shape[0]), dst.shape[1], dst.stride(0),
                  _ptr(src.amax), _stream())
        return
    _lib.call("radet_merge_planes", _ptr(src.t), C.c_void_p(dst.data_ptr()), C.c_size_t(dst.shape[0]), dst.shape[1],
              dst.stride(0), _stream())


def _is16(t):
    return t is not None and not _isp(t) and t.dtype == torch.bfloat16


def _tile(g, tile, default, x=None, y=None):
    """tile_override word: explicit or tuned tile + arithmetic mode flags, derived from the tensors' dtypes"""
    t = (tile or default) | (MATH_BF16 if g.math else 0)
    if _isp(x):                                          # plane operands: 0x200 (the fp32 paths' K-step bit) has no meaning here
        if (t & 0xFF) == 8:                              # (tile 8 splits in registers: fp32 operands only; tile 7 has a pair reader)
            t = (t & ~0xF0FF) | 3
        return (t & ~(MATH_BF16 | 0x200)) | P3 | (H2 if x.kind == "h2" else 0)
    if getattr(g, "x3", False) and not g.math and not _is16(x):
        t |= X3 | (H2 if getattr(g, "h2", False) else 0)
    if _is16(x):
        t = (t & ~MATH_BF16) | STORE_BF16 | (OUT_F32 if (y is not None and y.dtype == torch.float32) else 0)
    return t


def conv_fwd(g, x, wf, bias, y, addend=None, mask=None, relu=False, tile=0, splitk=True, yq=None, wmeta=None):
    """yq (Planes "h2", optional): y once more as fp16 plane pairs, scaled by the bound the launch can form before it starts
    (include/radet_hip.h RadetScales.yq); wmeta = (w_l1 slot, bias amax slot) of the conv, from the fold.  y = None with a
    yq: the output exists ONLY as pairs (round 6: tensors that nothing but conv GEMMs and ReLU masks read)"""
    assert y is not None or yq is not None
    tile = _tile(g, tile, g.fwd_tile, x, y)
    ws = splitk_ws() if splitk else None
    table = g.fwd_table
    sc = _scales(x, wf, y, need=bool(tile & H2), yq=yq, wmeta=wmeta, addend=addend)

    def launch():
        _lib.call("radet_conv2d_igemm_s", _ptr_any(x), _ptr_any(wf), _ptr(bias), _ptr(addend), _ptr(mask), _ptr(y), _ptr(table),
                  g.lout.rows, g.cin, g.cout, g.k, g.k, int(relu), tile, _ptr(ws), C.c_size_t(ws.numel() if splitk else 0),
                  _stream(), sc)
    _timed(lambda: _igemm_key(tile, x), 2.0 * g.lout.rows * g.cout * g.cin * g.k * g.k, launch, _conv_bytes(g), geom=g)


def _pred_tiles(lv):
    """8 x 16 output blocks of every (level, image) of the pyramid `lv` for radet_pred3x3_patch (built once per Levels)."""
    import numpy as np
    t = getattr(lv, "_pred_tiles", None)
    if t is None:
        rows = []
        for (h, w), off in zip(lv.hw, lv.offsets):
            for n in range(lv.B):
                for ty in range(-(-h // 8)):
                    for tx in range(-(-w // 16)):
                        rows.append((off + n * h * w, h, w, (ty << 16) | tx))
        t = torch.from_numpy(np.asarray(rows, np.int32)).cuda()
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream().synchronize()
        lv._pred_tiles = t
    return t


def pred_conv_patch(lv, x, a, b=None):
    """Predictor 3x3 convs from an LDS patch: a, b = (wf [c][9][Cin], bias, y [rows][c], c); b shares a's input."""
    t = _pred_tiles(lv)
    cin = x.shape[1]
    w1, b1, y1, c1 = b if b is not None else (None, None, None, 0)
    _timed("pred3x3_patch_kernel", 2.0 * lv.rows * (a[3] + c1) * cin * 9,
           lambda: _lib.call("radet_pred3x3_patch", _ptr(x), cin, _ptr(t), t.shape[0], _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), a[3],
                             _ptr(w1), _ptr(b1), _ptr(y1), c1, _stream()))


_WTUNE_CACHE = _LRU(8192)


def autotune_wgrad(g, reps=None):
    """Pick (tile, pixel splits S) of the one-tap wgrad kernel for this geometry: time the candidates and charge each
    split its downstream cost (a weight-sized slab is written here and read again by unfold: ~2 * 4 B / weight at
    ~3 TB/s).  Must run before the slab buffers are sized (it changes g.nsplit)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    reps = reps or TUNE_REPS
    global _TUNE_DIRTY, TUNE_RUNS
    key = (g._key, g.cin, g.cout, g.math, g.h16, "w") + (("x3",) if getattr(g, "x3", False) else ()) + _h2key(g)
    if key not in _WTUNE_CACHE:
        _TUNE_DIRTY = True
        TUNE_RUNS += 1
        M, kk = g.lout.rows, g.k * g.k
        s0 = g.nsplit
        cands = [(0, s0)]
        if g.cout > 64 and g.cin > 64:
            shapes = [(1 << 4, 128, 128), (2 << 4, 64, 64)]
            if getattr(g, "x3", False):
                shapes.append((3 << 4, 128, 64))          # plane arithmetic only: fewer operand splits per MFMA block
            for tflag, t, tn in shapes:
                tiles = -(-g.cout // t) * -(-g.cin // tn) * kk
                for blocks in (256, 512, 768, 1024):
                    S = max(1, min(64, round(blocks / tiles), (M + 127) // 128))
                    cands.append((tflag | 0x40, S))
                    if not g.math and not g.h16:
                        cands.append((tflag | 0x40 | 0x80, S))       # 32 pixels per stage
                    if getattr(g, "x3", False) and t == 64:
                        # the waves divide the pixels of a stage (64 four ways / 32 two ways) and share the operand splits
                        cands += [(tflag | 0x40 | 0x400, S), (tflag | 0x40 | 0x800, S)]
        cands = sorted(set(cands))
        dt = torch.bfloat16 if g.h16 else torch.float32
        dy = torch.randn(M, g.cout, device=dev).to(dt)
        x = torch.randn(g.lin.rows, g.cin, device=dev).to(dt)
        tmp_keys = _tune_slots(g, dy, x)
        slabs = torch.empty(max(c[1] for c in cands) * g.cout * kk * g.cin, device=dev)
        for fl, S in cands:
            g.wgrad_flags, g.nsplit = fl, S
            conv_wgrad(g, dy, x, slabs)
        torch.cuda.synchronize()
        tbest = [float("inf")] * len(cands)
        for _ in range(reps):
            for i, (fl, S) in enumerate(cands):
                g.wgrad_flags, g.nsplit = fl, S
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _b in range(TUNE_BURST):
                    conv_wgrad(g, dy, x, slabs)
                e.record()
                e.synchronize()
                tbest[i] = min(tbest[i], s.elapsed_time(e) / TUNE_BURST)
        best = None
        for (fl, S), t in zip(cands, tbest):
            cost = t + S * g.cout * kk * g.cin * 8 / 3e12 * 1e3
            if best is None or cost < best[0]:
                best = (cost, fl, S)
        _WTUNE_CACHE[key] = best[1:]
        unregister_amax(tmp_keys)
        if os.environ.get("RADET_TUNE_LOG"):
            print(f"[tune wgrad] M={M} {g.cin}->{g.cout} k{g.k}s{g.stride}: heuristic S={s0} -> flags={best[1]:#x} S={best[2]} "
                  f"({best[0] * 1e3:.1f} us incl. slab cost)")
    g.wgrad_flags, g.nsplit = _WTUNE_CACHE[key]


def conv_fwd_pair(g, a, b, relu=False, tile=0):
    """a, b: dicts(x, w, bias, addend, mask, y) -- two convs of geometry g in one launch."""
    ws = splitk_ws()
    q = lambda d: [_ptr_any(d.get(k)) for k in ("x", "w", "bias", "addend", "mask", "y")]  # noqa: E731
    t, table = _tile(g, tile, g.fwd_tile, a["x"], a["y"]), g.fwd_table
    sc = _scales(a["x"], a["w"], a["y"], b["x"], b["w"], b["y"], need=bool(t & H2))
    _timed(lambda: _igemm_key(t, a["x"]), 4.0 * g.lout.rows * g.cout * g.cin * g.k * g.k,
           lambda: _lib.call("radet_conv2d_igemm_pair_s", *q(a), *q(b), _ptr(table), g.lout.rows, g.cin, g.cout, g.k, g.k,
                             int(relu), t, _ptr(ws), C.c_size_t(ws.numel()), _stream(), sc), _conv_bytes(g, 2))


def conv_dgrad_pair(g, a, b, tile=0):
    """a, b: dicts(x=dy, w=wft, addend, mask, y=dx) -- two stride-1 dgrads of geometry g in one launch."""
    assert g.stride == 1
    ws = splitk_ws()
    q = lambda d: [_ptr_any(d.get(k)) for k in ("x", "w", "bias", "addend", "mask", "y")]  # noqa: E731
    t, table = _tile(g, tile, g.bwd_tile, a["x"], a["y"]), g.bwd_table
    sc = _scales(a["x"], a["w"], a["y"], b["x"], b["w"], b["y"], need=bool(t & H2))
    _timed(lambda: _igemm_key(t, a["x"]), 4.0 * g.lin.rows * g.cout * g.cin * g.k * g.k,
           lambda: _lib.call("radet_conv2d_igemm_pair_s", *q(a), *q(b), _ptr(table), g.lin.rows, g.cout, g.cin, g.k, g.k, 0,
                             t, _ptr(ws), C.c_size_t(ws.numel()), _stream(), sc), _conv_bytes(g, 2), kind="dgrad")


def conv_dgrad(g, dy, wft, dx, addend=None, mask=None, k_channels=None, tile=0, splitk=True, skip_zero_rows=False,
               yq=None, wmeta=None):
    """dx[rows_in, cin] = dgrad(dy[rows_out, k_channels]); k_channels = (padded) channel count of dy/wft.
    skip_zero_rows: the caller accumulates in place (addend is dx) with a mask already applied to dx, so the
    input positions a strided conv never touches (3/4 of them for a 1x1 / 2) are left alone instead of being
    rewritten by an epilogue-only launch.
    mask may be a Planes "h2" (the ReLU mask of an activation that exists only as pairs).  yq (Planes "h2") with wmeta =
    (w_l1t slot, None): dx as fp16 plane pairs scaled by amax(dy) * the weights' largest input-channel L1 norm; dx = None:
    only as pairs (stride-1 convs)."""
    kc = g.cout if k_channels is None else k_channels
    tile = _tile(g, tile, (g.bwd_tile_q if _isp(dy) and getattr(g, "bwd_tile_q", 0) else g.bwd_tile), dy, dx)
    if _isp(mask):
        tile |= MASKQ
    assert dx is not None or (yq is not None and g.stride == 1)
    ws = splitk_ws() if splitk else None
    if EVENTS is not None:
        # algorithmic work of the dgrad = that of the forward conv (a strided conv's dgrad touches each weight tap once
        # per output pixel; the padded K of the small predictor heads is not counted)
        real_k = min(kc, g.cout)
        _timed(_igemm_key(tile, dy) + (" [strided dgrad, class launch]" if g.stride > 1 else ""),
               2.0 * g.lout.rows * real_k * g.cin * g.k * g.k,
               lambda: _conv_dgrad(g, dy, wft, dx, addend, mask, kc, tile, ws, splitk, skip_zero_rows, yq, wmeta),
               _conv_bytes(g), kind="dgrad", geom=g)
        return
    _conv_dgrad(g, dy, wft, dx, addend, mask, kc, tile, ws, splitk, skip_zero_rows, yq, wmeta)


def _conv_dgrad(g, dy, wft, dx, addend, mask, kc, tile, ws, splitk, skip_zero_rows, yq=None, wmeta=None):
    # (a K that cannot be split into planes -- the 16-channel padded gradient of the reg / iou predictors -- runs on the
    # native fp32 MFMA inside the launcher and needs no operand slots)
    sc = _scales(dy, wft, dx, need=bool(tile & H2) and (_isp(dy) or kc % 32 == 0), yq=yq, wmeta=wmeta, addend=addend)
    if g.stride > 1 and STRIDED_DGRAD_CLASSES:
        assert yq is None and not _isp(mask) and not _isp(dy), "strided dgrad: fp32 tensors only"
        grp = _strided_dgrad_group(g)
        if grp is not None:
            _lib.call("radet_conv2d_igemm_classes_s", _ptr(dy), _ptr(wft), _ptr(addend), _ptr(mask), _ptr(dx),
                      _ptr(grp["table"]), _ptr(grp["out_rows"]), grp["tap_ids"], grp["ntaps"], grp["start"], grp["ncls"],
                      g.k * g.k, grp["M"], kc, g.cin, tile, _ptr(ws), C.c_size_t(ws.numel() if splitk else 0), _stream(), sc)
        for c in _strided_dgrad_classes(g):
            if grp is not None and not c["zero"]:
                continue
            if c["zero"] and skip_zero_rows:
                assert addend is not None and addend.data_ptr() == dx.data_ptr()
                continue
            # rows that receive no tap at all only need the epilogue: one 16-deep stage over an all-(-1) table (a K-divided
            # tile needs 32 / 64 channels per stage: the plain 64 x 64 tile instead, no split-K)
            ct = ((tile & ~0xF0FF) | 3) if (c["zero"] and (tile & 0xFF) >= 7) else tile
            _lib.call("radet_conv2d_igemm_taps_s", _ptr(dy), _ptr(wft), _ptr(addend), _ptr(mask), _ptr(dx), _ptr(c["table"]),
                      _ptr(c["out_rows"]), c["tap_ids"], c["ntaps"], g.k * g.k, c["rows"],
                      (32 if _is16(dy) else 16) if c["zero"] else kc, g.cin,
                      ct, _ptr(ws), C.c_size_t(ws.numel() if splitk else 0), _stream(), sc)
        return
    _lib.call("radet_conv2d_igemm_s", _ptr_any(dy), _ptr_any(wft), None, _ptr(addend), _ptr_any(mask), _ptr(dx), _ptr(g.bwd_table),
              g.lin.rows, kc, g.cin, g.k, g.k, 0, tile, _ptr(ws), C.c_size_t(ws.numel() if splitk else 0), _stream(), sc)


# all-taps weight gradient on fp16 plane pairs, unit-stride 3 x 3 convs: RADET_WGRAD9_WINDOWS=1 selects conv_wgrad9r_kernel (x as
# shifted windows of three row segments, every tile load a whole cache line: 15 instead of 26 KiB and 118 lines instead of 416
# half lines of LDS fill per 16 pixels; bit-identical results) -- built, tested, and SLOWER (167 against 107-117 us on the tower
# shape: neither the bytes nor the number of requests bound conv_wgrad9q_kernel), so off by default
WGRAD9_WINDOWS = os.environ.get("RADET_WGRAD9_WINDOWS", "0") == "1"
# ... RADET_WGRAD9_DEEP=1 selects conv_wgrad9d_kernel: conv_wgrad9q_kernel with five stage buffers, the loads four 16-pixel stages
# ahead, the gather table of the pixel split in LDS; bit-identical results.  4-9 % faster alone (91 against 95 us on the tower
# shape, 105 against 115 with bias sums), 0.3-0.5 % SLOWER in the step: its 159 KiB of LDS leave no room for a dgrad workgroup
# on the same CU (the two-buffer kernel: 104 KiB).  Off by default
WGRAD9_DEEP = os.environ.get("RADET_WGRAD9_DEEP", "0") == "1"


def _wgrad_key(g, dy, co):
    if _isp(dy):
        if dy.kind == "h2" and (g.k != 3 or g.wgrad_pair_flags & 0x40):
            return f"conv_wgradq_kernel<{'128, 128' if (g.wgrad_pair_flags >> 4) & 3 == 1 else '64, 64'}> (fp16 plane pairs)"
        if dy.kind == "h2":
            if g.stride == 1 and g.pad == 1 and WGRAD9_DEEP and -(-g.lout.rows // (16 * (g.nsplit_pairs or g.nsplit))) * 16 <= 1664:
                return "conv_wgrad9d_kernel"
            return "conv_wgrad9r_kernel" if (g.stride == 1 and g.pad == 1 and WGRAD9_WINDOWS) else "conv_wgrad9q_kernel"
        return "conv_wgrad9p_kernel"
    if getattr(g, "h2", False) and getattr(g, "x3", False) and not g.math and not _is16(dy):
        tf = (g.wgrad_flags >> 4) & 3
        tile = {0: "launcher tile", 1: "128, 128", 2: "64, 64", 3: "128, 64"}[tf] if co > 32 else "32, 128"
        kd = ", 64 px / 4 waves" if g.wgrad_flags & 0x400 else (", 32 px / 2 waves" if g.wgrad_flags & 0x800 else "")
        return f"conv_wgradg_kernel<{tile}> (fp16 hi/lo in registers{', 32 px' if g.wgrad_flags & 0x80 else ''}{kd})"
    nine = g.k == 3 and g.cin % 32 == 0 and co >= 256 and g.lout.rows >= 16384 and not (g.wgrad_flags & 0x40)
    mode = "bf16 storage" if _is16(dy) else ("bf16 math" if g.math else ("planes in registers" if getattr(g, "x3", False) else "fp32 MFMA"))
    if nine:
        return f"conv_wgrad9{'h' if _is16(dy) else 'g'}_kernel ({mode})"
    tf = (g.wgrad_flags >> 4) & 3
    tile = {0: "launcher tile", 1: "128, 128", 2: "64, 64", 3: "128, 64"}[tf] if co > 32 else "32, 128"
    kd = ", 64 px / 4 waves" if g.wgrad_flags & 0x400 else (", 32 px / 2 waves" if g.wgrad_flags & 0x800 else "")
    return f"conv_wgrad{'h' if _is16(dy) else 'g'}_kernel<{tile}> ({mode}{', 32 px' if g.wgrad_flags & 0x80 else ''}{kd})"


def conv_wgrad(g, dy, x, slabs, dbias_partials=None, cout=None, ld_dy=None):
    if EVENTS is not None:
        co_ = g.cout if cout is None else cout
        _timed(_wgrad_key(g, dy, co_), 2.0 * g.lout.rows * co_ * g.cin * g.k * g.k,
               lambda: _conv_wgrad(g, dy, x, slabs, dbias_partials, cout, ld_dy),
               4.0 * (g.lout.rows * co_ + g.lin.rows * g.cin + g.nsplit * co_ * g.k * g.k * g.cin), kind="wgrad", geom=g)
        return
    _conv_wgrad(g, dy, x, slabs, dbias_partials, cout, ld_dy)


def _conv_wgrad(g, dy, x, slabs, dbias_partials=None, cout=None, ld_dy=None):
    co = g.cout if cout is None else cout
    if _isp(dy):
        assert _isp(x) and x.kind == dy.kind
        if dy.kind == "h2":
            # all nine taps per workgroup for 3 x 3 convs unless the geometry's flags ask for the one-tap pair kernel (0x40;
            # bits 4-5 = its tile), which also serves every other kernel size
            nine = g.k == 3 and not (g.wgrad_pair_flags & 0x40)
            fl = 0x1000 | 0x200 | (g.wgrad_pair_flags if nine else (0x40 | (g.wgrad_pair_flags & 0x30)))
            if nine and g.stride == 1 and g.pad == 1:       # unit stride, padding 1 (what the two kernels below rely on)
                fl |= (0x2000 if WGRAD9_DEEP else 0) | (0x4000 if WGRAD9_WINDOWS else 0)
            _lib.call("radet_conv2d_wgrad_s", _ptr(dy.t), _ptr(x.t), _ptr(slabs), _ptr(dbias_partials), _ptr(g.fwd_table),
                      g.lout.rows, g.cin, co, co if ld_dy is None else ld_dy, g.k, g.k,
                      g.nsplit_pairs or g.nsplit, fl, _stream(), _scales(dy, x, None))
            return
        _lib.call("radet_conv2d_wgrad", _ptr(dy.t), _ptr(x.t), _ptr(slabs), _ptr(dbias_partials), _ptr(g.fwd_table), g.lout.rows,
                  g.cin, co, co if ld_dy is None else ld_dy, g.k, g.k, g.nsplit, 0x200 | (g.wgrad_flags & 0x40), _stream())
        return
    if getattr(g, "h2", False) and getattr(g, "x3", False) and not g.math and not _is16(dy):
        # fp16 hi / lo arithmetic on fp32 tensors: one-tap tiles only (an accumulator pair per tap does not fit the all-taps
        # tile); 0x40 keeps the launcher away from it for the geometries the tuner has not seen
        _lib.call("radet_conv2d_wgrad_s", _ptr(dy), _ptr(x), _ptr(slabs), _ptr(dbias_partials), _ptr(g.fwd_table), g.lout.rows,
                  g.cin, co, co if ld_dy is None else ld_dy, g.k, g.k, g.nsplit, 0x1000 | 0x40 | g.wgrad_flags, _stream(),
                  _scales(dy, x, None))
        return
    _lib.call("radet_conv2d_wgrad", _ptr(dy), _ptr(x), _ptr(slabs), _ptr(dbias_partials), _ptr(g.fwd_table), g.lout.rows,
              g.cin, co, co if ld_dy is None else ld_dy, g.k, g.k, g.nsplit,
              (2 if _is16(dy) else (1 if g.math else (0x100 if getattr(g, "x3", False) else 0))) | g.wgrad_flags, _stream())


def conv_wgrad_group(jobs, tile=128, math=0):
    """jobs: list of dict(g=ConvGeom, dy, x, slabs, dbias) -- one grouped launch per <= 32 jobs (longest first)."""
    jobs = sorted(jobs, key=lambda j: -(j["g"].lout.rows // j["g"].nsplit))
    for i in range(0, len(jobs), 32):
        part = jobs[i:i + 32]
        arr = (_lib.RadetWgradJob * len(part))()
        for a, j in zip(arr, part):
            g = j["g"]
            a.dy, a.x, a.slabs = _ptr(j["dy"]), _ptr(j["x"]), _ptr(j["slabs"])
            a.dbias_partials, a.gather_table = _ptr(j.get("dbias")), _ptr(g.fwd_table)
            a.M, a.Cin, a.Cout, a.ld_dy, a.KH, a.KW, a.S = g.lout.rows, g.cin, g.cout, g.cout, g.k, g.k, g.nsplit
        _lib.call("radet_conv2d_wgrad_group", arr, len(part), (1 if math else 0) | ((1 if tile == 128 else 2) << 4), _stream())


def group_splits(geoms, tile=128, slots=512):
    """Pixel splits S per conv of one wgrad group: S = round(M / px) with the pixel target px chosen so that the group's
    workgroup count fills `slots` resident workgroups (2 per CU) in whole rounds; ties -> fewer splits (fewer slabs)."""
    best = None
    for px in (6400, 4800, 3200, 2400, 1600, 1200, 800, 600, 400):
        S = [max(1, min(64, round(g.lout.rows / px), (g.lout.rows + 127) // 128)) for g in geoms]
        blocks = sum((g.cout // tile) * (g.cin // tile) * g.k * g.k * s for g, s in zip(geoms, S))
        eff = blocks / (slots * -(-blocks // slots))
        cost = -eff + 0.02 * sum(S) / len(S)            # every split is a slab written here and read by the reduction
        if best is None or cost < best[0] - 1e-9:
            best = (cost, S)
    return best[1]


def fold_weights(table_dev, n):
    _lib.call("radet_fold_weights", _ptr(table_dev), n, _stream())


def unfold_grads(table_dev, n, max_cout):
    _lib.call("radet_unfold_grads", _ptr(table_dev), n, max_cout, _stream())


def _h(name, t):
    """entry point for the tensor's storage type"""
    return name + "_h" if _is16(t) else name


def stem(img, wf, bias, y, B, H, W):
    ho, wo = (H + 1) // 2, (W + 1) // 2
    if _is16(y):
        fn = lambda: _lib.call("radet_stem_conv_bn_relu_h", _ptr(img), _ptr(wf), _ptr(bias), _ptr(y), B, H, W, _stream())  # noqa: E731
    else:
        sl = amax_slot(y)
        fn = lambda: _lib.call("radet_stem_conv_bn_relu_a", _ptr(img), _ptr(wf), _ptr(bias), _ptr(y), B, H, W, _ptr(sl), _stream())  # noqa: E731
    _timed("stem_kernel", 2.0 * B * ho * wo * 64 * 147, fn, 4.0 * (B * 3 * H * W + 64 * 147 + B * ho * wo * 64))


def convert_rows(src, dst, ncols=None, src_off=0, dst_off=0):
    """dst[:, dst_off:dst_off+ncols] = src[:, src_off:src_off+ncols] across fp32 <-> bf16 (2-D row-major tensors)"""
    assert src.dim() == 2 and dst.dim() == 2 and src.shape[0] == dst.shape[0] and src.dtype != dst.dtype
    n = src.shape[1] if ncols is None else ncols
    _lib.call("radet_convert_rows", _ptr(src), _ptr(dst), C.c_size_t(src.shape[0]), n, src.stride(0), src_off,
              dst.stride(0), dst_off, 1 if dst.dtype == torch.bfloat16 else 0, _stream())


def maxpool(x, y, B, H, W, Cch, yq=None):
    """yq (Planes "h2", optional): the output once more as plane pairs, scaled by x's amax slot"""
    if _is16(x):
        _lib.call("radet_maxpool3x3s2_h", _ptr(x), _ptr(y), B, H, W, Cch, _stream())
        return
    if yq is not None:
        _lib.call("radet_maxpool3x3s2_q", _ptr(x), _ptr(y), B, H, W, Cch, _ptr(amax_slot(y)), _ptr(yq.t), _ptr(yq.amax),
                  _ptr(amax_slot(x, True)), _stream())
        return
    _lib.call("radet_maxpool3x3s2_a", _ptr(x), _ptr(y), B, H, W, Cch, _ptr(amax_slot(y)), _stream())


def maxpool_bwd_relu(s, dpool, ds, B, H, W, Cch):
    """ds = [s > 0] * max-pool backward of dpool (s = the stem's ReLU output [B*H*W, C], fp32)"""
    _lib.call("radet_maxpool3x3s2_bwd_relu", _ptr(s), _ptr(dpool), _ptr(ds), B, H, W, Cch, _stream())


def stem_wgrad_splits(B, H, W):
    return _lib.load().radet_stem_wgrad_splits(B, H, W)


def stem_wgrad(img, ds, slabs, dbias_partials, B, H, W, S):
    """slabs[S][64][49][3], dbias_partials[S][64] from the NCHW image and the stem's pre-activation gradient"""
    _lib.call("radet_stem_wgrad", _ptr(img), _ptr(ds), _ptr(slabs), _ptr(dbias_partials), B, H, W, S, _stream())


def gn_ws_floats(levels):
    d, n = _desc([(h, w, h, w, o, o) for (h, w), o in zip(levels.hw, levels.offsets)])
    return _lib.load().radet_gn_workspace_floats(levels.B, d, n)


def _gn_desc(levels):
    return _desc([(h, w, h, w, o, o) for (h, w), o in zip(levels.hw, levels.offsets)])


def gn_relu_fwd(levels, z, gamma, beta, y, stats, ws, eps=1e-5, relu=True):
    d, n = _gn_desc(levels)
    _lib.call(_h("radet_gn_relu_fwd", z), _ptr(z), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(stats), _ptr(ws), levels.B, 256, 32,
              eps, int(relu), d, n, _stream())


def gn_relu_fwd_pair(levels, a, b, eps=1e-5, relu=True):
    """GroupNorm + ReLU of two tensors of one geometry in one pair of launches; a / b = (z, gamma, beta, y, stats, ws)."""
    d, n = _gn_desc(levels)
    _lib.call(_h("radet_gn_relu_fwd_pair", a[0]), *[_ptr(t) for t in a], *[_ptr(t) for t in b], levels.B, 256, 32, eps,
              int(relu), d, n, _stream())


def gn_relu_fwd_p(levels, z, gamma, beta, y, yp, stats, ws, eps=1e-5, relu=True):
    """fp32 z -> y (fp32 tensor or None) and / or yp (Planes or None)"""
    d, n = _gn_desc(levels)
    _lib.call("radet_gn_relu_fwd_p", _ptr(z), _ptr(gamma), _ptr(beta), _ptr(y), _ptr_any(yp), _ptr(stats), _ptr(ws), levels.B,
              256, 32, eps, int(relu), d, n, _stream())


def gn_relu_fwd_pair_p(levels, a, b, eps=1e-5, relu=True):
    """a / b = (z, gamma, beta, y or None, yp (Planes) or None, stats, ws)"""
    d, n = _gn_desc(levels)
    _lib.call("radet_gn_relu_fwd_pair_p", *[_ptr_any(t) for t in a], *[_ptr_any(t) for t in b], levels.B, 256, 32, eps,
              int(relu), d, n, _stream())


def gn_relu_fwd_q(levels, z, gamma, beta, y, yq, stats, ws, zhat_amax=None, eps=1e-5, relu=True):
    """fp16 hi / lo arithmetic: fp32 z -> y (fp32 tensor or None; its registered amax slot is raised) and / or yq (Planes of
    kind "h2" or None: scaled by a bound on |y|, written to yq.amax); zhat_amax: slot raised to the largest |zhat|"""
    d, n = _gn_desc(levels)
    _lib.call("radet_gn_relu_fwd_q", _ptr(z), _ptr(gamma), _ptr(beta), _ptr(y), _ptr_any(yq), _ptr(stats), _ptr(ws), levels.B,
              256, 32, eps, int(relu), d, n, _stream(), _ptr(amax_slot(y)), _ptr(yq.amax) if yq is not None else None,
              _ptr(zhat_amax))


def gn_relu_fwd_pair_q(levels, a, b, eps=1e-5, relu=True):
    """a / b = (z, gamma, beta, y or None, yq (Planes "h2") or None, stats, ws, zhat_amax or None)"""
    d, n = _gn_desc(levels)

    def q(t):
        z, gm, bt, y, yq, stats, ws, zh = t
        return [_ptr(z), _ptr(gm), _ptr(bt), _ptr(y), _ptr_any(yq), _ptr(stats), _ptr(ws), _ptr(amax_slot(y)),
                _ptr(yq.amax) if yq is not None else None, _ptr(zh)]
    _lib.call("radet_gn_relu_fwd_pair_q", *q(a), *q(b), levels.B, 256, 32, eps, int(relu), d, n, _stream())


def gn_relu_bwd_q(levels, dy, z, stats, gamma, beta, dz, dzq, dgamma, dbeta, ws, zhat_amax=None, dy_amax=None, relu=True):
    """like gn_relu_bwd with dz as fp32 (or None) and / or Planes "h2" dzq (scaled by a bound built from dy's amax slot --
    dy_amax, else the registered one, else computed here -- and zhat_amax; the bound goes to dzq.amax)"""
    d, n = _gn_desc(levels)
    da = dy_amax if dy_amax is not None else (amax_slot(dy, compute=True) if dzq is not None else None)
    _lib.call("radet_gn_relu_bwd_q", _ptr(dy), _ptr(z), _ptr(stats), _ptr(gamma), _ptr(beta), _ptr(dz), _ptr_any(dzq),
              _ptr(dgamma), _ptr(dbeta), _ptr(ws), levels.B, 256, 32, int(relu), d, n, _stream(), _ptr(da), _ptr(zhat_amax),
              _ptr(dzq.amax) if dzq is not None else None)


def gn_relu_bwd_p(levels, dy, z, stats, gamma, beta, dz, dzp, dgamma, dbeta, ws, relu=True):
    """like gn_relu_bwd with dz as fp32 (or None) and / or Planes dzp (or None)"""
    d, n = _gn_desc(levels)
    _lib.call("radet_gn_relu_bwd_p", _ptr(dy), _ptr(z), _ptr(stats), _ptr(gamma), _ptr(beta), _ptr(dz), _ptr_any(dzp),
              _ptr(dgamma), _ptr(dbeta), _ptr(ws), levels.B, 256, 32, int(relu), d, n, _stream())


def gn_relu_bwd(levels, dy, z, stats, gamma, beta, dz, dgamma, dbeta, ws, relu=True):
    d, n = _gn_desc(levels)
    _lib.call(_h("radet_gn_relu_bwd", dy), _ptr(dy), _ptr(z), _ptr(stats), _ptr(gamma), _ptr(beta), _ptr(dz), _ptr(dgamma),
              _ptr(dbeta), _ptr(ws), levels.B, 256, 32, int(relu), d, n, _stream())


def upsample_add(dst, src, B, ho, wo, hi, wi, ch):
    if _is16(dst):
        _lib.call("radet_upsample_add_h", _ptr(dst), _ptr(src), B, ho, wo, hi, wi, ch, _stream())
        return
    _lib.call("radet_upsample_add_a", _ptr(dst), _ptr(src), B, ho, wo, hi, wi, ch, _ptr(amax_slot(dst)), _stream())


def upsample_add_bwd(dsrc, ddst, B, ho, wo, hi, wi, ch):
    if _is16(dsrc):
        _lib.call("radet_upsample_add_bwd_h", _ptr(dsrc), _ptr(ddst), B, ho, wo, hi, wi, ch, _stream())
        return
    _lib.call("radet_upsample_add_bwd_a", _ptr(dsrc), _ptr(ddst), B, ho, wo, hi, wi, ch, _ptr(amax_slot(dsrc)), _stream())


def relu_bwd(dy, addend, act, dx):
    if _is16(dy):
        _lib.call("radet_relu_bwd_h", _ptr(dy), _ptr(addend), _ptr(act), _ptr(dx), C.c_size_t(dx.numel()), _stream())
        return
    _lib.call("radet_relu_bwd_a", _ptr(dy), _ptr(addend), _ptr(act), _ptr(dx), C.c_size_t(dx.numel()), _ptr(amax_slot(dx)),
              _stream())


def nchw_to_nhwc(x, y, B, ch, H, W):
    """module-API boundary (fp32 NCHW in); a bf16 row buffer is filled through an fp32 staging tensor"""
    if _is16(y):
        tmp = torch.empty(y.shape, device=y.device, dtype=torch.float32)
        _lib.call("radet_nchw_to_nhwc", _ptr(x), _ptr(tmp), B, ch, H, W, _stream())
        convert_rows(tmp.view(-1, ch), y.view(-1, ch))
        return
    _lib.call("radet_nchw_to_nhwc", _ptr(x), _ptr(y), B, ch, H, W, _stream())
    sl = amax_slot(y)
    if sl is not None and y.numel() % 4 == 0:          # data entering an engine buffer from outside: its amax slot follows
        absmax(y, sl)


def nhwc_to_nchw(x, y, B, ch, H, W):
    """module-API boundary (fp32 NCHW out)"""
    if _is16(x):
        tmp = torch.empty(x.shape, device=x.device, dtype=torch.float32)
        convert_rows(x.reshape(-1, ch), tmp.view(-1, ch))
        x = tmp
    _lib.call("radet_nhwc_to_nchw", _ptr(x), _ptr(y), B, ch, H, W, _stream())


def level_desc(levels, strides):
    flat = []
    for (h, w), s in zip(levels.hw, strides):
        flat += [h, w, int(s)]
    return (C.c_int * len(flat))(*flat), len(levels.hw)


def head_loss_ws_ints(R):
    return _lib.load().radet_head_loss_ws_ints(R)


def head_loss(cls, reg_u, iou, scales, gt_boxes, gt_labels, gt_off, p2g, pw, ldesc, nlvl, B, num_classes, alpha, gamma,
              lbw, giou_eps, grad_scale, losses, dcls, dcls_ld, dreg, dreg_ld, diou, diou_ld, dscales, ws,
              labels_out=None, tgt_out=None, flags=0):
    _lib.call("radet_head_loss", _ptr(cls), _ptr(reg_u), _ptr(iou), _ptr(scales), _ptr(gt_boxes), _ptr(gt_labels),
              _ptr(gt_off), _ptr(p2g), _ptr(pw), ldesc, nlvl, B, num_classes, alpha, gamma, lbw, giou_eps,
              _ptr(grad_scale), _ptr(losses), _ptr(dcls), dcls_ld, _ptr(dreg), dreg_ld, _ptr(diou), diou_ld,
              _ptr(dscales), _ptr(labels_out), _ptr(tgt_out), flags, _ptr(ws), _stream())


def scale_relu(reg_u, scales, out, ldesc, nlvl, B):
    _lib.call("radet_scale_relu", _ptr(reg_u), _ptr(scales), _ptr(out), ldesc, nlvl, B, _stream())


def grid_anchors(out, ldesc, nlvl, base_scale=8):
    _lib.call("radet_grid_anchors", _ptr(out), ldesc, nlvl, base_scale, _stream())


# ---------------------------------------------------------------------- stand-alone box / loss operators
OVERLAP_MODES = dict(iou=0, iof=1, giou=2)


def bbox_overlaps(b1, b2, out, batch, M, N, mode, aligned, eps):
    _lib.call("radet_bbox_overlaps", _ptr(b1), _ptr(b2), _ptr(out), batch, M, N, OVERLAP_MODES[mode], int(aligned), eps,
              _stream())


def _norm4(normalizer):
    v = [float(normalizer)] * 4 if isinstance(normalizer, (int, float)) else [float(x) for x in normalizer]
    assert len(v) == 4, "Normalizer must have length = 4"
    return (C.c_float * 4)(*v)


def tblr_encode(priors, gts, out, normalizer, normalize_by_wh=True):
    _lib.call("radet_tblr_encode", _ptr(priors), _ptr(gts), _ptr(out), priors.shape[0], _norm4(normalizer),
              int(normalize_by_wh), _stream())


def tblr_decode(priors, tblr, out, normalizer, normalize_by_wh=True, max_shape=None, clip_border=True):
    clip = bool(clip_border and max_shape is not None)
    mh, mw = (float(max_shape[0]), float(max_shape[1])) if clip else (0.0, 0.0)
    _lib.call("radet_tblr_decode", _ptr(priors), _ptr(tblr), _ptr(out), priors.shape[0], _norm4(normalizer),
              int(normalize_by_wh), mh, mw, int(clip), _stream())


def loss_partials(n_elem):
    return _lib.load().radet_loss_partials(n_elem)


def loss_finalize(partials, avg_factor, scale, out):
    _lib.call("radet_loss_finalize", _ptr(partials), partials.numel(), _ptr(avg_factor), scale, _ptr(out), _stream())


def sigmoid_focal_loss(x, target, weight, wcols, N, Cc, gamma, alpha, elem, partials, elem_scale=1.0):
    _lib.call("radet_sigmoid_focal_loss", _ptr(x), _ptr(target), _ptr(weight), wcols, C.c_size_t(N), Cc, gamma, alpha,
              _ptr(elem), elem_scale, _ptr(partials), _stream())


def sigmoid_focal_loss_bwd(x, target, weight, wcols, N, Cc, gamma, alpha, grad_elem, grad_scalar, avg_factor, scale, dx):
    _lib.call("radet_sigmoid_focal_loss_bwd", _ptr(x), _ptr(target), _ptr(weight), wcols, C.c_size_t(N), Cc, gamma, alpha,
              _ptr(grad_elem), _ptr(grad_scalar), _ptr(avg_factor), scale, _ptr(dx), _stream())


def bce_logits_loss(x, target, weight, wcols, N, Cc, elem, partials, elem_scale=1.0):
    _lib.call("radet_bce_logits_loss", _ptr(x), _ptr(target), _ptr(weight), wcols, C.c_size_t(N), Cc, _ptr(elem),
              elem_scale, _ptr(partials), _stream())


def bce_logits_loss_bwd(x, target, weight, wcols, N, Cc, grad_elem, grad_scalar, avg_factor, scale, dx):
    _lib.call("radet_bce_logits_loss_bwd", _ptr(x), _ptr(target), _ptr(weight), wcols, C.c_size_t(N), Cc, _ptr(grad_elem),
              _ptr(grad_scalar), _ptr(avg_factor), scale, _ptr(dx), _stream())


def giou_loss(pred, target, weight, N, eps, elem, partials, elem_scale=1.0):
    _lib.call("radet_giou_loss", _ptr(pred), _ptr(target), _ptr(weight), C.c_size_t(N), eps, _ptr(elem), elem_scale, _ptr(partials),
              _stream())


def giou_loss_bwd(pred, target, weight, N, eps, grad_elem, grad_scalar, avg_factor, scale, dpred):
    _lib.call("radet_giou_loss_bwd", _ptr(pred), _ptr(target), _ptr(weight), C.c_size_t(N), eps, _ptr(grad_elem),
              _ptr(grad_scalar), _ptr(avg_factor), scale, _ptr(dpred), _stream())


def threshold_compact(scores, thr, idx, count):
    _lib.call("radet_threshold_compact", _ptr(scores), C.c_size_t(scores.numel()), thr, _ptr(idx), _ptr(count), _stream())


def sqnorm_partials(g, n, partials):
    _lib.call("radet_sqnorm_partials", _ptr(g), C.c_size_t(n), _ptr(partials), partials.numel(), _stream())


def adamw_step(p, g, m, v, n, lr, betas, eps, wd, step, max_norm, grad_div, partials, grad_norm_out):
    _lib.call("radet_adamw_step", _ptr(p), _ptr(g), _ptr(m), _ptr(v), C.c_size_t(n), lr, betas[0], betas[1], eps, wd,
              step, max_norm, grad_div, _ptr(partials), partials.numel(), _ptr(grad_norm_out), _stream())


def decode_ws_bytes(B, nlvl, nms_pre):
    return int(_lib.load().radet_decode_ws_bytes(B, nlvl, nms_pre))


def decode_candidates(cls, reg_u, iou, scales, ldesc, nlvl, B, num_classes, score_thr, nms_pre, img_hw, scale_factor,
                      cand_boxes, cand_scores, cand_ctr, cand_labels, cand_count, ws):
    _lib.call("radet_decode_candidates", _ptr(cls), _ptr(reg_u), _ptr(iou), _ptr(scales), ldesc, nlvl, B, num_classes,
              score_thr, nms_pre, _ptr(img_hw), _ptr(scale_factor), _ptr(cand_boxes), _ptr(cand_scores), _ptr(cand_ctr),
              _ptr(cand_labels), _ptr(cand_count), _ptr(ws), _stream())


def nms_ws_bytes(B, cap):
    return int(_lib.load().radet_nms_ws_bytes(B, cap))


NMS_MODES = dict(vote=0, global_vote=1, cluster=2, nms=3)


def nms(boxes, cluster_scores, vote_scores, labels, counts, B, cap, mode, iou_thr, iou_enable, sigma, max_out, out_boxes,
        out_scores, out_labels, out_count, aux0, aux1, ws):
    _lib.call("radet_nms", _ptr(boxes), _ptr(cluster_scores), _ptr(vote_scores), _ptr(labels), _ptr(counts), B, cap, mode,
              iou_thr, int(iou_enable), sigma, max_out, _ptr(out_boxes), _ptr(out_scores), _ptr(out_labels),
              _ptr(out_count), _ptr(aux0), _ptr(aux1), _ptr(ws), _stream())


def mbd_ws_bytes(px):
    return int(_lib.load().radet_mbd_ws_bytes(px))


def mbd(images, desc, n, sx, sy, alpha, niter, base_size, dmap, px, ws):
    _lib.call("radet_mbd", _ptr(images), _ptr(desc), n, _ptr(sx), _ptr(sy), alpha, niter, base_size, _ptr(dmap),
              C.c_size_t(px), _ptr(ws), _stream())


def gdt(cost, desc, n, sx, sy, dist, ws):
    _lib.call("radet_gdt", _ptr(cost), _ptr(desc), n, _ptr(sx), _ptr(sy), _ptr(dist), _ptr(ws), _stream())


def assign_ws_bytes(B, N):
    return int(_lib.load().radet_assign_ws_bytes(B, N))


FLIP = {None: 0, "none": 0, "horizontal": 1, "vertical": 2, "diagonal": 3}


def mask_transform(src, out_hw=None, resized_hw=None, flip=None, pad_val=0, normalize=False):
    """src: u8[G,Hs,Ws] device tensor -> u8[G,Hd,Wd]: nearest resize to resized_hw, flip, pad to out_hw (one pass)."""
    G, Hs, Ws = src.shape
    Hr, Wr = resized_hw or (Hs, Ws)
    Hd, Wd = out_hw or (Hr, Wr)
    dst = torch.empty(G, Hd, Wd, dtype=torch.uint8, device=src.device)
    if G == 0:
        return dst
    mx = None
    if normalize:
        mx = torch.empty(G, dtype=torch.int32, device=src.device)
        _lib.call("radet_mask_max", _ptr(src), _ptr(mx), G, C.c_size_t(Hs * Ws), _stream())
    _lib.call("radet_mask_transform", _ptr(src), _ptr(dst), _ptr(mx), G, Hs, Ws, Hr, Wr, Hd, Wd, FLIP[flip], int(pad_val),
              _stream())
    return dst


def assign_points(gt_boxes, gt_off, masks, H, W, rng_words, U, ldesc, ranges, nlvl, B, positive_num, neg_thr, p2g, pw, used,
                  ws, flags=1):
    """masks: u8 [sumG, H, W] visible masks, or f32 per-box distance maps (mask-free sampler); rng_words: int32 / uint32 bit
    patterns [B, U] of the RandomStates' next raw outputs; flags: bit 0 balance_sample, bit 1 multiply_samplepro_for_weight,
    bit 2 adapt_positive_num, bit 3 uniform integer draws (random_sample_by_distance=False)"""
    _lib.call("radet_assign_points_f" if masks.dtype == torch.float32 else "radet_assign_points", _ptr(gt_boxes), _ptr(gt_off),
              _ptr(masks), H, W, _ptr(rng_words), U, ldesc, ranges, nlvl, B, positive_num, int(flags), neg_thr, _ptr(p2g), _ptr(pw),
              _ptr(used), _ptr(ws), _stream())


# ---------------------------------------------------------------------- image processing around MBD / GDT (packed crops)
def gauss9_kernel():
    """cv::getGaussianKernel(9, sigma <= 0 -> 0.3 * ((9 - 1) * 0.5 - 1) + 0.8 = 1.7, CV_32F): float taps, normalised with a
    double sum; returned as (centre, +-1, +-2, +-3, +-4)"""
    import numpy as np
    sigma = 0.3 * ((9 - 1) * 0.5 - 1) + 0.8
    x = np.arange(9, dtype=np.float64) - 4.0
    cf = np.exp(-0.5 / (sigma * sigma) * x * x).astype(np.float32)
    s = 0.0
    for v in cf:
        s += float(v)
    cf = (cf.astype(np.float64) * (1.0 / s)).astype(np.float32)
    return (C.c_float * 5)(*[float(v) for v in cf[4:]])


def resize_linear_u8(src, sdesc, dst, ddesc, n, max_dst_px, channels=3):
    _lib.call("radet_resize_linear_u8", _ptr(src), _ptr(sdesc), _ptr(dst), _ptr(ddesc), n, max_dst_px, channels, _stream())


def resize_linear_f(src, sdesc, dst, ddesc, n, max_dst_px):
    _lib.call("radet_resize_linear_f", _ptr(src), _ptr(sdesc), _ptr(dst), _ptr(ddesc), n, max_dst_px,
              1 if src.dtype == torch.float64 else 0, _stream())


def gaussian_blur9_u8(src, desc, dst, tmp, n, max_px):
    _lib.call("radet_gaussian_blur9_u8", _ptr(src), _ptr(desc), _ptr(dst), _ptr(tmp), gauss9_kernel(), n, max_px, _stream())


def sobel_edge(src, desc, edge, gray_ws, max_ws, n, max_px):
    _lib.call("radet_sobel_edge", _ptr(src), _ptr(desc), _ptr(edge), _ptr(gray_ws), _ptr(max_ws), n, max_px, _stream())
