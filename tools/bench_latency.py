"""Single-image inference latency (BASELINE config 4 at B=1): eager launches vs one hipGraph replay."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.argv = [sys.argv[0]]
from tools import bench_configs as bc

B = int(os.environ.get("B", "1"))
cfg, det = bc.build(50)
det.eval()
rt = det.runtime()
g = torch.Generator().manual_seed(0)
imgs = torch.randn(B, 3, 480, 640, generator=g).cuda()
metas = [dict(img_shape=(480, 640, 3), scale_factor=np.ones(4, np.float32)) for _ in range(B)]
rt.detect(imgs, metas, det.test_cfg, rescale=True)
with torch.no_grad():
    logits = rt.engine.buf["cls"].flatten()
    q = torch.quantile(logits[:2_000_000].float(), 0.98)
    det.bbox_head.atss_cls.bias += float(np.log(0.05 / 0.95)) - float(q)
for _ in range(5):
    rt.detect(imgs, metas, det.test_cfg, rescale=True)
torch.cuda.synchronize()
n = 100
t0 = time.perf_counter()
for _ in range(n):
    out = rt.detect(imgs, metas, det.test_cfg, rescale=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
# host-only enqueue time: no sync between calls except detect's own result fetch
print(f"eager  B={B}: {dt * 1e3:.2f} ms per call ({B / dt:.0f} images/s)")
if hasattr(rt, "detect_graph") and os.environ.get("LAT_NO_GRAPH") != "1":
    for _ in range(3):
        out2 = rt.detect_graph(imgs, metas, det.test_cfg, rescale=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out2 = rt.detect_graph(imgs, metas, det.test_cfg, rescale=True)
    torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t0) / n
    same = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(out, out2))
    print(f"graph  B={B}: {dt2 * 1e3:.2f} ms per call ({B / dt2:.0f} images/s), results identical to eager: {same}")
    if not same:
        for a, b in zip(out, out2):
            print("  eager", tuple(a[0].shape), "graph", tuple(b[0].shape),
                  "max |d box|", float((a[0][:min(len(a[0]), len(b[0]))] - b[0][:min(len(a[0]), len(b[0]))]).abs().max()) if len(a[0]) and len(b[0]) else None,
                  "labels equal", bool((a[1][:min(len(a[1]), len(b[1]))] == b[1][:min(len(a[1]), len(b[1]))]).all()))
