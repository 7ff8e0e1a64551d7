"""Round 6: is there anything left for a weights-resident STREAMING kernel on the large-M / small-K 1 x 1 convs of layer1 / layer2?
Per forward launch of those stages IN the train step (HIP events around every conv launch, radet_amd.kernels.EVENTS): duration,
the bytes the launch has to move at least (operand pairs in, fp32 + pair copy out, the residual it adds) and the rate that makes.

    python tools/stream_check.py      -> profiles/round6_stream_check.txt
"""
import os
import sys

os.environ["RADET_TAPE"] = "0"                       # (per-launch events need the eager step)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from radet_amd import kernels as K  # noqa: E402
from radet_amd.models import build_detector  # noqa: E402
from radet_amd.utils import Config  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    for _ in range(4):
        rt.train_step(img, tg)
    torch.cuda.synchronize()
    steps = 10
    K.EVENTS = ev = []
    for _ in range(steps):
        rt.train_step(img, tg)
    torch.cuda.synchronize()
    K.EVENTS = None
    agg = {}
    order = []
    for x in ev:
        if x["kind"] != "fwd" or x["stage"] not in ("layer1", "layer2") or x.get("geom") is None:
            continue
        g = x["geom"]
        key = (x["stage"], g.lin.rows, g.lout.rows, g.cin, g.cout, g.k, g.stride, x["key"])
        if key not in agg:
            agg[key] = [0, 0.0]
            order.append(key)
        agg[key][0] += 1
        agg[key][1] += x["start"].elapsed_time(x["end"]) * 1e3
    print("forward launches of layer1 / layer2 in the train step (B = 4, 640 x 480; default fp16 hi / lo arithmetic).  `least bytes`:")
    print("input as pairs (4 B per element; a 3 x 3 reads it once) + weights + output as fp32 AND pairs (8 B) -- the residual of")
    print("a block's last conv (4 B per output element more) is NOT counted, so that column understates those launches.")
    tot_us = tot_b = 0.0
    for key in order:
        stage, rin, rout, cin, cout, k, s, name = key
        n, us = agg[key]
        per = n / steps
        us /= n
        rows_read = rin if s == 1 else rout * (k * k if k > 1 else 1)          # (a strided 1 x 1 reads every s^2-th row)
        rows_read = min(rows_read, rin)
        b = 4.0 * rows_read * cin + 4.0 * cout * k * k * cin + 8.0 * rout * cout
        tot_us += us * per
        tot_b += b * per
        print(f"{stage} rows {rin:6d}->{rout:6d} {cin:4d}->{cout:4d} k{k}s{s} x{per:3.0f}  {us:6.1f} us  least {b / 1e6:6.1f} MB  "
              f"{b / us / 1e6:5.2f} TB/s  {2.0 * rout * cin * cout * k * k / us / 1e6:6.1f} TFLOP/s   {name}")
    print(f"sum per step: {tot_us:.0f} us for at least {tot_b / 1e6:.0f} MB = {tot_b / tot_us / 1e6:.2f} TB/s")


if __name__ == "__main__":
    main()
