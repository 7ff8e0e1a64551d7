"""What a stream of the USER's costs next to the engine's four (DESIGN.md 5 "stream budget"):
the headline train step (a) alone, (b) with a user-created HIP stream that uploads the next batch (pinned H2D copy of a
4 x 3 x 480 x 640 image tensor per step, what a data loader's copy stream does), (c) the same upload on one of the engine's own streams (the tower-chain stream, idle outside the head), (d) with a user stream
that exists but is never used, (e) the upload on the main stream in front of the step.  python tools/bench_user_stream.py [steps]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from radet_amd.models import build_detector
from radet_amd.utils import Config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg = Config.fromfile(os.path.join(ROOT, "configs", "bop", "r50_ycbv_pbr.py")); cfg.model["pretrained"] = None
torch.manual_seed(0)
det = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
rt = det.runtime(); rt.init_optimizer()
dev = torch.device("cuda")
img, boxes, labels, p2g, pw = bench.make_batch(0, 4, dev)
tg = rt.pack_targets([torch.from_numpy(b) for b in boxes], [torch.from_numpy(l) for l in labels], list(p2g), list(pw))
host = img.cpu().pin_memory()
nxt = torch.empty_like(img)


def run(copy_stream=None, label=""):
    for _ in range(5):
        rt.train_step(img, tg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        if copy_stream is not None:
            with torch.cuda.stream(copy_stream):
                nxt.copy_(host, non_blocking=True)
        rt.train_step(img, tg)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{label:70s} {ms:7.3f} ms/step", flush=True)
    return ms


a = run(None, "(a) train step alone (engine: main + 3 streams)")
c = run(rt.engine._chain_stream(), "(c) + next-batch upload on the engine's tower-chain stream")
a2 = run(None, "(a) again")
user = torch.cuda.Stream()
d = run(None, "(d) a fifth stream exists, never used")
b = run(user, "(b) + next-batch upload on a user-created fifth stream")
b2 = run(None, "(a) after the fifth stream was used")
e = run(torch.cuda.current_stream(), "(e) + next-batch upload on the main stream, in front of the step")
print(f"main-stream upload {e / a2:.3f}")
print(f"ratios vs (a): chain stream {c / a:.3f}, idle fifth stream {d / a2:.3f}, busy fifth stream {b / a2:.3f}, afterwards {b2 / a2:.3f}")
# Round 5 (tools/micro/stream_cliff.hip): HIP hands hardware queues to streams in the order of their FIRST USE, and queues k and
# k + 4 share a pipe of the command processor.  The engine's four streams hold queues 0-3, so a fifth stream gets queue 4 = the
# MAIN stream's pipe.  Touching placeholder streams first moves the upload to queue 5 / 6 / 7 = the pipe of another engine stream.
g = run(rt.engine.caller_stream(), "(g) upload on Engine.caller_stream()")
run(None, "(a) afterwards")
ph = []
for n in (1, 2, 3):
    p = torch.cuda.Stream()
    with torch.cuda.stream(p):
        torch.zeros(1, device=dev)
    ph.append(p)
    u = torch.cuda.Stream()
    f = run(u, f"(f{n}) upload on a stream first used after {n} placeholder stream(s)")
    run(None, "(a) afterwards")
