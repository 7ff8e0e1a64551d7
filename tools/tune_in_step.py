"""In-step re-tuning of the dgrad tiles of the backbone / neck (round 6).

The start-up tuner (radet_amd/kernels.py:autotune) times every launch ALONE on an idle device.  That is the right yardstick for
the forward pass, whose launches do run alone, but not for the backward pass: its dgrad chain on the main stream shares the CUs
with the weight gradients of two more streams, and a tile that wins alone can lose there (round 6: the bf16 towers' dgrad,
64 x 64 alone, 128 x 128 in the step: -4.4 % step time).  This tool measures where it matters: for every distinct dgrad geometry
of the trainable backbone / neck it runs the REAL train step with each candidate tile and HIP events around every conv launch
(radet_amd.kernels.EVENTS) and scores a candidate by the summed in-step duration of the main stream's dgrad launches of the
backbone + neck (the critical path of that phase).  The winners are verified against the incumbent tune on the step time itself
(alternating blocks of steps) and written to a tune file (same format as radet_amd/tune_gfx950.json).

    python tools/tune_in_step.py gpurun_out/tune_instep.json [--math fp32|bf16-storage] [--batch 4] [--steps 6]
"""
import argparse
import json
import os
import sys
import time

os.environ["RADET_TAPE"] = "0"                       # (per-launch events need the eager step)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from radet_amd import kernels as K  # noqa: E402
from radet_amd.models import build_detector  # noqa: E402
from radet_amd.utils import Config  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tune_key(g):
    return (g._key, g.cin, g.cout, g.math, g.h16) + (("x3",) if getattr(g, "x3", False) else ()) + K._h2key(g)


def candidates(g):
    """the start-up tuner's dgrad candidates for geometry g (kernels.autotune: cands(g.cout, g.cin, g.lin.rows, taps))"""
    kdim, n, m, taps = g.cout, g.cin, g.lin.rows, g.k * g.k
    tiles = [4] if n <= 32 else [1, 2, 3]
    c = list(tiles)
    if kdim % 32 == 0:
        c += [t | 0x200 for t in tiles]
    if getattr(g, "x3", False) and not g.math and not g.h16 and n > 32:
        c += ([7] if kdim % 64 == 0 else []) + ([8] if kdim % 32 == 0 else [])
    out = list(c)
    for t in c:
        bm = 64 if (t & 0xFF) in (3, 7, 8) else 128
        bn = {1: 128, 2: 64, 3: 64, 4: 32, 7: 64, 8: 64}[t & 0xFF]
        ntiles = -(-m // bm) * -(-n // bn)
        nk = taps * kdim // (64 if (t & 0xFF) == 7 else (32 if (t & 0x200 or (t & 0xFF) == 8) else 16))
        if ntiles < 1024:
            out += [t | (sk << 12) for sk in (1, 2, 3, 4, 6, 8) if nk // sk >= 4]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--math", default="fp32")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--verify-steps", type=int, default=40)
    args = ap.parse_args()
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "bop", "r50_ycbv_pbr.py"))
    cfg.model["pretrained"] = None
    torch.manual_seed(0)
    det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
    rt = det.runtime(math=args.math)
    rt.init_optimizer()
    rt.set_loss_from_head(det.bbox_head)
    img, boxes, labels, p2g, pw = bench.make_batch(0, args.batch, torch.device("cuda"))
    tg = rt.pack_targets([torch.from_numpy(b) for b in boxes], [torch.from_numpy(l) for l in labels], list(p2g), list(pw))
    for _ in range(4):
        rt.train_step(img, tg)
    torch.cuda.synchronize()
    e = rt.engine
    groups = {}                                            # tune key -> geometries (one per conv) that share it
    for c in e.convs:
        if c.trainable and c.need_dgrad and c.geom is not None and c.name.startswith(("backbone.", "neck.")) and c.geom.cout % 16 == 0:
            groups.setdefault(tune_key(c.geom), []).append(c.geom)
    chain_stages = ("layer1", "layer2", "layer3", "layer4", "neck")

    def measure(steps):
        """(summed in-step duration of the backbone / neck dgrad launches per step, {group key: its launches' share}, wall ms per step)"""
        torch.cuda.synchronize()
        K.EVENTS = ev = []
        t0 = time.perf_counter()
        for _ in range(steps):
            rt.train_step(img, tg)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps * 1e3
        K.EVENTS = None
        tot, per = 0.0, {}
        for x in ev:
            if x["kind"] == "dgrad" and x["stage"] in chain_stages:
                d = x["start"].elapsed_time(x["end"])
                tot += d
                if x.get("geom") is not None:
                    k = tune_key(x["geom"])
                    per[k] = per.get(k, 0.0) + d
        return tot / steps, {k: v / steps for k, v in per.items()}, wall

    base_chain, base_per, base_wall = measure(args.steps)
    print(f"incumbent: backbone + neck dgrad chain {base_chain * 1e3:.0f} us in the step, {base_wall:.3f} ms per (instrumented) step, "
          f"{len(groups)} dgrad geometries", flush=True)
    picks = {}
    for key, geoms in sorted(groups.items(), key=lambda kv: -base_per.get(kv[0], 0.0)):
        g0 = geoms[0]
        cur = g0.bwd_tile
        cands = [cur] + [t for t in candidates(g0) if t != cur]
        best = None
        rows = []
        for t in cands:
            for g in geoms:
                g.bwd_tile = t
            try:
                chain, per, _ = measure(args.steps)
            except Exception as ex:                           # a tile the launcher refuses for this shape
                rows.append((t, None, None))
                K.EVENTS = None
                print("   tile %#x refused: %s" % (t, str(ex)[:80]), flush=True)
                continue
            own = per.get(key, 0.0)
            rows.append((t, chain, own))
            if best is None or chain < best[1]:
                best = (t, chain, own)
        # a challenger has to beat the incumbent's chain time by more than the measurement's scatter (two more looks at both)
        t_new = best[0]
        if t_new != cur:
            a = b = 0.0
            for _ in range(2):
                for g in geoms:
                    g.bwd_tile = cur
                a += measure(args.steps)[0]
                for g in geoms:
                    g.bwd_tile = t_new
                b += measure(args.steps)[0]
            if b > a * 0.997:
                t_new = cur
        for g in geoms:
            g.bwd_tile = t_new
        picks[key] = t_new
        inc = [r for r in rows if r[0] == cur][0]
        print(f"M={g0.lin.rows:6d} {g0.cout:4d}->{g0.cin:4d} k{g0.k}s{g0.stride} x{len(geoms)}: incumbent {cur:#x} own {inc[2] * 1e3:6.1f} us chain "
              f"{inc[1] * 1e3:6.0f} | best {best[0]:#x} own {best[2] * 1e3:6.1f} chain {best[1] * 1e3:6.0f} -> {t_new:#x}", flush=True)
    # verification on the step time itself: incumbent against the new picks, alternating blocks
    def set_tiles(new):
        for key, geoms in groups.items():
            for g in geoms:
                g.bwd_tile = picks[key] if new else K._TUNE_CACHE[key][1]

    def step_ms(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            rt.train_step(img, tg)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    res = {False: [], True: []}
    for rep in range(4):
        for new in (False, True):
            set_tiles(new)
            step_ms(5)
            res[new].append(step_ms(args.verify_steps))
    old_ms, new_ms = min(res[False]), min(res[True])
    print(f"step time: incumbent tiles {old_ms:.3f} ms (runs {[round(v, 3) for v in res[False]]}), in-step tiles {new_ms:.3f} ms "
          f"(runs {[round(v, 3) for v in res[True]]})", flush=True)
    changed = {k: v for k, v in picks.items() if v != K._TUNE_CACHE[k][1]}
    print(f"{len(changed)} of {len(picks)} dgrad geometries change their tile", flush=True)
    if new_ms < old_ms * 0.997:
        for k, v in changed.items():
            K._TUNE_CACHE[k] = (K._TUNE_CACHE[k][0], v)
        K.save_tune_cache(args.out)
        print("written:", args.out)
    else:
        print("no gain beyond the scatter: nothing written")
        with open(args.out + ".rejected", "w") as fh:
            json.dump({repr(k): v for k, v in changed.items()}, fh, indent=0)


if __name__ == "__main__":
    main()
