"""Run-to-run comparison of the train step (round 6): the same model, the same batch, N optimisation steps, several times in
one process -- after every step EVERY engine buffer (plane tensors incl. their amax / true-amax slots), the weight-gradient
slabs, the folded weights, the gradient and parameter arenas are compared bit for bit with the first run's, and for a
differing activation the pattern of the difference is printed (which 64-row tiles, which 32-column blocks, how large).
This is what found the lo-fragment race of the plane-pair GEMM: one wave's 32 x 32 block off by 4e-5 in a few
tiles of a few runs.  Benign differences it reports: WHICH of a slot's 64 words holds the maximum (the last arriver of an
in-launch split-K reduction raises it), buffers nobody writes.     python tools/run_to_run_diff.py
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from radet_amd import kernels as K
from radet_amd.models import build_detector
from radet_amd.utils import Config
img, boxes, labels, p2g, pw = bench.make_batch(0, 4, torch.device("cuda"))

def run(nsteps):
    cfg = Config.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "bop", "r50_ycbv_pbr.py")); cfg.model["pretrained"] = None
    torch.manual_seed(0)
    det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
    rt = det.runtime(); rt.tape_mode = "0"; rt.init_optimizer(); rt.set_loss_from_head(det.bbox_head)
    tg = rt.pack_targets([torch.from_numpy(b) for b in boxes], [torch.from_numpy(l) for l in labels], list(p2g), list(pw))
    snaps = []
    for i in range(nsteps):
        rt.train_step(img, tg); torch.cuda.synchronize()
        e = rt.engine
        s = {}
        for k, v in e.buf.items():
            if K._isp(v):
                s[k] = v.t.clone(); s[k + "#amax"] = v.amax.clone()
                if hasattr(v, "true_amax") and v.true_amax is not None: s[k + "#true"] = v.true_amax.clone()
            elif torch.is_tensor(v):
                s[k] = v.clone()
        s["~grads"] = rt.flat.grads.clone(); s["~params"] = rt.flat.params.clone(); s["~slabs"] = e.slab_arena.clone()
        s["~amax_act"] = e.amax_act.clone(); s["~w_amax"] = e.w_amax.clone()
        if e.w_l1t is not None: s["~w_l1t"] = e.w_l1t.clone(); s["~w_l1"] = e.w_l1.clone(); s["~b_amax"] = e.b_amax.clone()
        s["~wf"] = e.wf_arena.clone(); s["~wft"] = e.wft_arena.clone()
        for c in e.convs:
            if K._isp(c.wf): s["~wfP." + c.name] = c.wf.t.clone(); s["~wftP." + c.name] = c.wft.t.clone()
        snaps.append(s)
    return snaps, rt

N = 4
ref, _ = run(N)
for rep in range(4):
    cur, rt = run(N)
    for i in range(N):
        bad = []
        for k in ref[i]:
            a, b = ref[i][k], cur[i][k]
            same = torch.equal(a.view(torch.uint8), b.view(torch.uint8)) if a.dtype == torch.float16 else torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a, b.view(torch.int32) if b.dtype == torch.float32 else b)
            if not same: bad.append(k)
        print("rep", rep, "step", i, "differing:", len(bad), bad[:14], flush=True)
        for k in bad:
            if k.endswith(".o2") or k.endswith(".o1") or k.endswith("d_o1"):
                a, b = ref[i][k].float(), cur[i][k].float()
                d = (a - b).abs()
                idx = (d > 0).nonzero()
                rows, cols = idx[:, 0], idx[:, 1]
                print("   ", k, tuple(a.shape), "n diff", idx.shape[0], "max abs", d.max().item(), "max val", a.abs().max().item(),
                      "row tiles(64)", sorted(set((rows // 64).tolist()))[:12], "n row tiles", len(set((rows // 64).tolist())),
                      "col tiles(32)", sorted(set((cols // 32).tolist()))[:12], "rows in tile", sorted(set((rows % 64).tolist()))[:20])
        if bad: break
