"""MBD / GDT batch timing: N crops of 150x200 (the reference resizes box crops to a 150-pixel short edge)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from radet_amd import ops
from oracle import dist
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rs = np.random.RandomState(0)
imgs = [(rs.rand(150, 200, 3) * 255).astype(np.uint8) for _ in range(N)]
costs = [rs.rand(150, 200).astype(np.float32) for _ in range(N)]
seeds = [dist.border_seeds(150, 200)] * N
for name, fn, data in (("MBD niter=4", lambda: ops.mbd_batch(imgs, seeds, 0.1, 4, 300), imgs), ("GDT", lambda: ops.gdt_batch(costs, seeds), costs)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    gpu = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    if name.startswith("MBD"): dist.mbd(imgs[0], *seeds[0], 0.1, 4, 300)
    else: dist.gdt(costs[0], *seeds[0])
    cpu = time.perf_counter() - t0
    print(f"{name}: {N} crops {gpu * 1e3:.2f} ms on the GPU (incl. host packing + upload) vs {cpu * 1e3:.2f} ms per crop on one CPU core (oracle)")
