"""What would the gradient exchange of an 8-GPU node cost the (throughput-bound) backward pass?  One GPU, no second rank: at
each bucket's hand-over point (runtime.backward's bucket hook, on the side stream) a stand-in kernel is put on the tower-chain
stream -- where the real collectives run -- that keeps W workgroups resident for the time an 8-rank ring all-reduce of the bucket
needs at an assumed bus bandwidth and moves a rank's share of the ring through HBM (tools/micro/comm_standin.hip).  Reports the
step time without exchange, with fp32 buckets and with bf16 buckets, W = 16 / 32 workgroups.
    python tools/comm_emulation.py [busbw GB/s, default 300] > profiles/round5_comm_emulation.txt
No scaling curve exists for this repository (no 8-GPU node was available to any round): this is an estimate of the TAX, not a
measurement of RCCL."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from radet_amd.models import build_detector
from radet_amd.utils import Config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "tools", "_probe", "libcomm_standin.so"))
lib.comm_standin.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_void_p]
busbw = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0


def main():
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "bop", "r50_ycbv_pbr.py"))
    cfg.model["pretrained"] = None
    torch.manual_seed(0)
    det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
    rt = det.runtime()
    rt.init_optimizer()
    rt.set_loss_from_head(det.bbox_head)
    img, boxes, labels, p2g, pw = bench.make_batch(0, 4, torch.device("cuda"))
    tg = rt.pack_targets([torch.from_numpy(b) for b in boxes], [torch.from_numpy(l) for l in labels], list(p2g), list(pw))
    e = rt.engine
    scratch = torch.empty(2 * 64 * 1024 * 1024, dtype=torch.uint8, device="cuda")        # 2 x 64 MiB
    log = []

    def make_hook(wgs, bytes_per_param):
        def hook(bucket):
            lo, hi = bucket["arena"]
            nbytes = (hi - lo) * bytes_per_param
            ring = int(2 * 7 / 8 * nbytes)                       # what one rank sends (and receives) in a ring all-reduce
            us = ring / (busbw * 1e9) * 1e6
            cs = e._chain_stream()
            ev = e._event()
            ev.record()                                          # (on the side stream: the bucket's slabs are reduced)
            cs.wait_event(ev)
            lib.comm_standin(scratch.data_ptr(), min(ring, scratch.numel() // 2) // 16 * 16, wgs, us, cs.cuda_stream)
            log.append((bucket["prefix"], nbytes, us))
        return hook

    def step(hook):
        rt.forward(img)
        rt.loss(tg, grad_scale=rt.loss_weights)
        rt.backward(hook)
        if hook is not None:
            e._join(e._chain_stream())                           # GradReducer.finish(): clip + AdamW wait for the exchange
        rt.optimizer_step(grad_div=1.0)

    def timeit(hook, n=30):
        for _ in range(5):
            step(hook)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step(hook)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    base = timeit(None)
    print(f"assumed bus bandwidth of the 8-rank ring: {busbw:.0f} GB/s (a rank sends and receives 2 * 7/8 of every bucket)")
    print(f"step without exchange: {base:.3f} ms")
    for wgs in (16, 32):
        for name, bpp in (("fp32 buckets", 4), ("bf16 buckets", 2)):
            log.clear()
            t = timeit(make_hook(wgs, bpp))
            per = {}
            for p, nb, us in log:
                per[p] = (nb, us)
            tot = sum(us for _, us in per.values())
            print(f"stand-in on the chain stream, {wgs} workgroups, {name}: {t:.3f} ms (+{t - base:.3f} ms, +{100 * (t / base - 1):.1f} %); "
                  f"buckets {', '.join(f'{p} {nb / 1e6:.1f} MB {us:.0f} us' for p, (nb, us) in per.items())}; exchange time summed {tot / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
