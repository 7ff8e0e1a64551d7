import os, sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_configs as t
from oracle import model as om, synth
from radet_amd.apis import wrap_fp16_model
H, W = 224, 224
img, *_ = t.batch(H, W, 2)
d = lambda a, b: (a - b).norm().item() / max(b.norm().item(), 1e-12)
det = t.make(50); wrap_fp16_model(det); det.eval()
with torch.no_grad():
    gC = [f.cpu().double() for f in det.backbone(img.cuda())]
odet = om.OracleDetector(50, seed=1)
variants = {}
def run(rx, rw):
    class V(torch.autograd.Function):
        pass
    def conv(x, w, stride, padding):
        return F.conv2d(om._r(x) if rx else x, om._r(w) if rw else w, None, stride=stride, padding=padding)
    orig = om._ConvBF16.apply
    om._ConvBF16.apply = staticmethod(lambda x, w, s, p: conv(x, w, s, p))
    try:
        with torch.no_grad(), om.conv_math("bf16"):
            return [f.double() for f in om.backbone(odet.sd, img, 50)]
    finally:
        om._ConvBF16.apply = orig
with torch.no_grad():
    ref32 = [f.double() for f in om.backbone(odet.sd, img, 50)]
for name, (rx, rw) in dict(both=(1, 1), x_only=(1, 0), w_only=(0, 1), none=(0, 0)).items():
    v = run(rx, rw)
    print(name, " ".join(f"C{l+2}: gpu-v {d(gC[l], v[l]):.2e} v-fp32 {d(v[l], ref32[l]):.2e} |" for l in range(4)))
