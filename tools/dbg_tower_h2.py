"""Ablation of the plane-pair tower GEMM (fp16 hi / lo, 256 x 128 8-wave tile) on the B = 4 tower shape: the launch with its tile
loads / MFMAs / fragment reads switched off one at a time (library built with -DRADET_P3_DBG=1: RADET_LIB=libradet_hip_dbg.so,
RADET_DBG_IGEMM bits 1 = no tile loads after the prologue, 2 = no MFMAs, 4 = no fragment reads after the first).
    for d in 0 1 2 4 3 5 6; do RADET_LIB=libradet_hip_dbg.so RADET_DBG_IGEMM=$d python tools/dbg_tower_h2.py; done"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radet_amd import kernels as K
from radet_amd.kernels import ConvGeom, Levels, Planes
from tools.bench_p3 import timeit
dev = torch.device("cuda")
lv = Levels([(60, 80), (30, 40), (15, 20), (8, 10), (4, 5)], 4)
g = ConvGeom(lv, 256, 256, 3, 1, 1); g.x3 = "h2"
x = torch.relu(torch.randn(lv.rows, 256, device=dev)); w = torch.randn(256 * 9, 256, device=dev) * 0.05
xp, wp = Planes.from_float(x, kind="h2"), Planes.from_float(w, kind="h2")
y = torch.empty(lv.rows, 256, device=dev)
flop = 2.0 * lv.rows * 256 * 256 * 9
for t in [6 | (1 << 12), 6 | (1 << 12) | K.ROWPAIRS, 5 | (1 << 12)]:
    us = timeit(lambda: K.conv_fwd(g, xp, wp, None, y, tile=t))
    print(f"dbg={os.environ.get('RADET_DBG_IGEMM','0'):>2s} tile {t:#x}: {us:8.1f} us {flop/us/1e6:7.1f} TFLOP/s (fp32-equivalent)", flush=True)
