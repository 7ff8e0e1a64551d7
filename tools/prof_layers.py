"""Per-launch timing of every implicit-GEMM / wgrad call of one train step (events around each call, serialised):
shape, microseconds, TFLOP/s.  python tools/prof_layers.py [fwd|bwd|wgrad]"""
import os, sys, collections
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from radet_amd import kernels as K
from radet_amd.models import build_detector
from radet_amd.utils import Config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RADET_TOWER_MODE"] = "serial"
cfg = Config.fromfile(os.path.join(ROOT, "configs", "bop", "r50_ycbv_pbr.py")); cfg.model["pretrained"] = None
torch.manual_seed(0)
det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
rt = det.runtime(); rt.init_optimizer()
rt.engine.use_streams = False
img, boxes, labels, p2g, pw = bench.make_batch(0, 4, torch.device("cuda"))
tg = rt.pack_targets([torch.from_numpy(b) for b in boxes], [torch.from_numpy(l) for l in labels], list(p2g), list(pw))
for _ in range(2):
    rt.train_step(img, tg)
rec = []
def wrap(name, kind):
    orig = getattr(K, name)
    def f(g, *a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); r = orig(g, *a, **k); e.record()
        kc = k.get("k_channels") or (k.get("cout") if kind == "wgrad" else None)
        rec.append((kind, g, s, e))
        return r
    setattr(K, name, f)
for n, kind in (("conv_fwd", "fwd"), ("conv_dgrad", "bwd"), ("conv_wgrad", "wgrad")):
    wrap(n, kind)
rt.train_step(img, tg)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for kind, g, s, e in rec:
    M = g.lout.rows if kind != "bwd" else g.lin.rows
    key = (kind, g.lin.rows, g.cin, g.cout, g.k, g.stride)
    fl = 2.0 * g.lout.rows * g.cin * g.cout * g.k * g.k
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1; a[1] += s.elapsed_time(e) * 1e3; a[2] += fl
want = sys.argv[1] if len(sys.argv) > 1 else None
tot = collections.Counter()
for (kind, rows, cin, cout, k, st), (n, us, fl) in agg.items():
    tot[kind] += us
    if want and kind != want: continue
    print(f"{kind:5s} rows_in={rows:6d} {cin:4d}->{cout:4d} k{k} s{st}  x{n:2d}  {us / n:8.1f} us  {fl / us / 1e6:6.1f} TF   total {us:8.1f} us")
print({k: round(v) for k, v in tot.items()})
