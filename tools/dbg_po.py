"""Debug aid: the pairs-only step (Engine.po) against the all-fp32-tensor step on the same batch -- first buffer / gradient
that differs by more than rounding.   python tools/dbg_po.py [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from radet_amd import kernels as K  # noqa: E402
from radet_amd.models import build_detector  # noqa: E402
from radet_amd.utils import Config  # noqa: E402


def build(po):
    os.environ["RADET_PAIRS_ONLY"] = po
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "bop", "r50_ycbv_pbr.py"))
    cfg.model["pretrained"] = None
    torch.manual_seed(0)
    det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
    if os.environ.get("DBG_SYNTH") == "1":
        from radet_amd.utils.synth_init import synth_fill
        synth_fill(det, seed=0)
    rt = det.runtime()
    rt.tape_mode = "0"
    rt.set_loss_from_head(det.bbox_head)
    return det, rt


def f32(t):
    return t.to_float() if K._isp(t) else t


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    img, boxes, labels, p2g, pw = bench.make_batch(0, B, torch.device("cuda"))
    res = {}
    for po in ("0", "1"):
        det, rt = build(po)
        tg = rt.pack_targets([torch.from_numpy(b) for b in boxes], [torch.from_numpy(l) for l in labels], list(p2g), list(pw))
        rt.forward(img)
        losses = rt.loss(tg).clone()
        torch.cuda.synchronize()
        e = rt.engine
        fw = {k: f32(v).clone() for k, v in e.buf.items() if (k.startswith("l") and k[1].isdigit() and not k.split(".")[-1].startswith("d_"))
              or k in ("P", "cls", "reg_u", "iou")}
        rt.backward()
        torch.cuda.synchronize()
        bw = {k: f32(v).clone() for k, v in e.buf.items() if k.split(".")[-1].startswith("d_") or k.startswith("d_")}
        res[po] = (losses.cpu(), fw, bw, rt.flat.grads.clone(), {n: (o, rt.flat.p[n].numel()) for n, o in rt.flat.offsets.items()})
        print("po", po, "losses", losses.cpu().tolist(), "blocks po:", [[blk["po"] for blk in st] for st in e.stages])
    (l0, f0, b0, g0, offs), (l1, f1, b1, g1, _) = res["0"], res["1"]

    def cmp(a, b):
        d = (a.double() - b.double()).abs().max().item()
        return d / max(a.double().abs().max().item(), 1e-30), torch.isfinite(b).all().item()
    for name, (d0, d1) in (("forward", (f0, f1)), ("backward", (b0, b1))):
        print("==", name)
        for k in d0:
            if k in d1 and d0[k].shape == d1[k].shape:
                r, fin = cmp(d0[k], d1[k])
                if r > 1e-5 or not fin:
                    print(f"  {k:16s} rel {r:.3e} finite {fin} max0 {d0[k].abs().max().item():.3e} max1 {d1[k].abs().max().item():.3e}")
    print("== parameter gradients")
    bad = 0
    for n, (o, cnt) in offs.items():
        r, fin = cmp(g0[o:o + cnt], g1[o:o + cnt])
        if r > 1e-4 or not fin:
            bad += 1
            if bad < 40:
                print(f"  {n:50s} rel {r:.3e} finite {fin}")
    print("bad gradient tensors:", bad, "of", len(offs))


if __name__ == "__main__":
    main()
