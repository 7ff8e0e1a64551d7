"""ORACLE ctypes wrapper over oracle/liboracle.so (dist.c): MBD / GDT box-to-distance transforms and the seed layout of
the reference wrapper (radet/ops/bbox2distance/bbox2distance_wrapper.py:20-39).  Test infrastructure only."""
import ctypes

import numpy as np

from . import nms as _n


def border_seeds(h, w, interval=3):
    """seeds on the image border every `interval` pixels: top, bottom, left, right (bbox2distance_wrapper.py:22-36)"""
    hx = list(range(0, w, interval))
    if hx[-1] != w - 1:
        hx.append(w - 1)
    hx = np.asarray(hx, np.int64)
    vy = np.arange(1, h - 1, interval, dtype=np.int64)
    sx = np.concatenate([hx, hx, np.zeros_like(vy), np.full_like(vy, w - 1)])
    sy = np.concatenate([np.zeros_like(hx), np.full_like(hx, h - 1), vy, vy])
    return sx, sy


def mbd(image, seeds_x, seeds_y, alpha=0.1, niter=4, base_size=300):
    image = np.ascontiguousarray(image, np.uint8)
    h, w, _ = image.shape
    sx, sy = np.ascontiguousarray(seeds_x, np.int64), np.ascontiguousarray(seeds_y, np.int64)
    out = np.empty((h, w), np.float64)
    _n.lib().oracle_mbd(_n._p(image), h, w, _n._p(sx), _n._p(sy), len(sx), ctypes.c_float(alpha), niter, base_size, _n._p(out))
    return out


def gdt(cost, seeds_x, seeds_y):
    cost = np.ascontiguousarray(cost, np.float32)
    h, w = cost.shape
    sx, sy = np.ascontiguousarray(seeds_x, np.int64), np.ascontiguousarray(seeds_y, np.int64)
    out = np.empty((h, w), np.float32)
    _n.lib().oracle_gdt(_n._p(cost), h, w, _n._p(sx), _n._p(sy), len(sx), _n._p(out))
    return out
