"""Launch tape: the host side of a steady-state train step recorded once and replayed by ONE C call per segment
(include/radet_hip.h "launch tape", csrc/tape.hip).

Why: a train step is ~250 C-ABI calls on four HIP streams; issued from Python (argument marshalling through ctypes, slot /
table look-ups, stream context managers, event objects) they cost ~19 us each -- 4.5-4.9 ms of host time per step, which is
what bounds the bf16-storage step (5.3 ms) and what every step would be bound by on a node whose 8 ranks share the host.
The engine's step is a FIXED program over pre-allocated buffers (engine.py), so the calls of step k + 1 equal those of step
k except for a handful of argument words: the batch's pointers, the learning rate, the step number.

How: while `_lib.TAPE` is set, `_lib.call` appends every successful call (entry point + raw argument words) and
`kernels.ev_record / ev_wait` append the cross-stream event operations.  `cut()` closes a segment where Python has to run
between two launches (the gradient exchange: torch.distributed collectives).  `replay()` patches the per-step words and
hands each segment to radet_tape_replay -- the same calls, in the same order, on the same streams, so the results are
bit-identical to the eager step (tests/test_gpu_model.py::test_tape_replay_is_bit_identical).

What a tape must not see: an allocation during the recorded step (a temporary's address would be baked in and reused after
it is freed).  `begin()` / `end()` compare the caching allocator's allocation counter and poison the tape if it moved.
"""
import ctypes as C
import struct

import torch

from . import _lib

_KIND_CALL, _KIND_RECORD, _KIND_WAIT = 0, 1, 2


def _fbits(v):
    return struct.unpack("<I", struct.pack("<f", v))[0]


def _raw(a, ty):
    """the 64-bit argument word ctypes would pass for `a` under argtype `ty` (+ an object to keep alive, or None)"""
    if a is None:
        return 0, None
    if ty is C.c_float:
        v = a.value if isinstance(a, C._SimpleCData) else a
        return _fbits(v), None
    if isinstance(a, int):
        return a & 0xFFFFFFFFFFFFFFFF, None
    if isinstance(a, C._SimpleCData):
        v = a.value
        if isinstance(v, bytes):                      # c_char_p
            return C.cast(a, C.c_void_p).value or 0, a
        return (v or 0) & 0xFFFFFFFFFFFFFFFF, None
    if isinstance(a, (C.Array, C.Structure)):
        return C.addressof(a), a
    obj = getattr(a, "_obj", None)                    # byref(struct)
    if obj is not None:
        return C.addressof(obj), a
    raise TypeError(f"tape: cannot record an argument of type {type(a)}")


class TapeError(_lib.RadetHipError):
    pass


class Tape:
    def __init__(self):
        self._ops = []               # (kind, fn, stream, event, [arg words])
        self._keep = []              # host objects whose addresses the ops hold (ctypes arrays / structs, torch events)
        self._cuts = []              # (op index, callback, stream or None): Python to run between two segments
        self._marks = {}             # tag -> [(op index, arg index, is_float)]
        self._binds = {}             # name -> (last pointer, [(op index, arg index, byte offset)])
        self.poisoned = None         # reason why this tape must not be replayed
        self.ops = None              # the ctypes array, after end()
        self._alloc0 = None
        self.replays = 0

    # ------------------------------------------------------------------ recording
    @staticmethod
    def _alloc_count(dev):
        return torch.cuda.memory_stats(dev).get("allocation.all.allocated", 0)

    def begin(self, dev):
        assert _lib.TAPE is None, "a tape is already being recorded"
        self.dev = dev
        self._alloc0 = self._alloc_count(dev)
        _lib.TAPE = self
        return self

    def end(self):
        assert _lib.TAPE is self
        _lib.TAPE = None
        if self._alloc_count(self.dev) != self._alloc0:
            self.poisoned = "device memory was allocated while the step was recorded (a temporary's address would be replayed)"
        n = len(self._ops)
        arr = (_lib.RadetTapeOp * max(n, 1))()
        for o, (kind, fn, stream, event, words) in zip(arr, self._ops):
            o.kind, o.fn, o.stream, o.event = kind, fn, stream, event
            for i, w in enumerate(words):
                o.args[i] = w
        self.ops = arr
        self.n = n
        self._segments = []
        lo = 0
        for idx, cb, st in self._cuts:
            self._segments.append((lo, idx, cb, st))
            lo = idx
        self._segments.append((lo, n, None, None))
        return self

    def abort(self):
        if _lib.TAPE is self:
            _lib.TAPE = None
        self.poisoned = "recording aborted"

    def record_call(self, name, args):
        res, argtypes = _lib.SIGNATURES[name]
        fn = _lib.load().radet_tape_fn_index(name.encode())
        if fn < 0 or len(args) != len(argtypes):
            self.poisoned = f"{name} cannot be taped"
            return
        words = []
        for a, ty in zip(args, argtypes):
            w, keep = _raw(a, ty)
            words.append(w)
            if keep is not None:
                self._keep.append(keep)
        self._ops.append((_KIND_CALL, fn, None, None, words))
        self._last_call = (len(self._ops) - 1, argtypes)

    def record_event(self, ev, stream_handle):
        self._keep.append(ev)
        self._ops.append((_KIND_RECORD, 0, stream_handle, ev.cuda_event, ()))

    def wait_event(self, ev, stream_handle):
        self._keep.append(ev)
        self._ops.append((_KIND_WAIT, 0, stream_handle, ev.cuda_event, ()))

    def mark(self, tag, arg):
        """argument `arg` of the call recorded last changes from step to step: replay(values={tag: v}) patches it"""
        idx, argtypes = self._last_call
        self._marks.setdefault(tag, []).append((idx, arg, argtypes[arg] is C.c_float))

    def cut(self, callback, stream=None):
        """Python runs here between two replayed segments: callback() with `stream` (a torch stream) current"""
        self._cuts.append((len(self._ops), callback, stream))

    def bind(self, name, t):
        """every pointer argument that points into tensor `t` follows the tensor handed to replay(tensors={name: ...})"""
        lo = t.data_ptr()
        hi = lo + max(t.numel() * t.element_size(), 1)
        sites = []
        for i, (kind, fn, _, _, words) in enumerate(self._ops):
            if kind != _KIND_CALL:
                continue
            argtypes = self._argtypes_of(fn)
            for j, (w, ty) in enumerate(zip(words, argtypes)):
                if ty is C.c_void_p and lo <= w < hi:
                    sites.append((i, j, w - lo))
        self._binds[name] = [lo, sites, (tuple(t.shape), t.dtype)]
        return len(sites)

    _ARGT = {}

    def _argtypes_of(self, fn):
        at = Tape._ARGT.get(fn)
        if at is None:
            lib = _lib.load()
            for name, (res, args) in _lib.SIGNATURES.items():
                k = lib.radet_tape_fn_index(name.encode())
                if k >= 0:
                    Tape._ARGT[k] = args
            at = Tape._ARGT[fn]
        return at

    # ------------------------------------------------------------------ replay
    def replay(self, tensors=None, values=None):
        if self.poisoned:
            raise TapeError(f"tape unusable: {self.poisoned}")
        ops = self.ops
        for name, t in (tensors or {}).items():
            b = self._binds[name]
            p = t.data_ptr()
            if p != b[0]:
                for i, j, off in b[1]:
                    ops[i].args[j] = p + off
                b[0] = p
        for tag, v in (values or {}).items():
            for i, j, is_f in self._marks.get(tag, ()):
                ops[i].args[j] = _fbits(v) if is_f else int(v) & 0xFFFFFFFFFFFFFFFF
        run = _lib.load().radet_tape_replay
        failed = C.c_int(-1)
        for lo, hi, cb, st in self._segments:
            if hi > lo:
                rc = run(ops, lo, hi, C.byref(failed))
                if rc != 0:
                    raise TapeError(f"tape op {failed.value} of {self.n} failed with code {rc}")
            if cb is not None:
                if st is not None:
                    with torch.cuda.stream(st):
                        cb()
                else:
                    cb()
        self.replays += 1

    def stats(self):
        calls = sum(1 for o in self._ops if o[0] == _KIND_CALL)
        return dict(ops=len(self._ops), calls=calls, events=len(self._ops) - calls, segments=len(self._segments),
                    bound={k: len(v[1]) for k, v in self._binds.items()})
