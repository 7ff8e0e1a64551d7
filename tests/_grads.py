"""Shared gradient comparisons of the parity tests (norms alone would survive a tap transposition or a sign error inside
a tensor: these compare the tensors themselves)."""
import numpy as np


def grad_rel_errors(mine, ref):
    """mine / ref: dict name -> gradient tensor.  Returns (per-parameter ||a - b|| / ||b||, total norm of ref)."""
    tot = float(np.sqrt(sum(float(g.double().norm()) ** 2 for g in ref.values())))
    out = {}
    for n, b in ref.items():
        a = mine[n].double().cpu().reshape(-1)
        b = b.double().cpu().reshape(-1)
        out[n] = (float((a - b).norm()), float(b.norm()))
    return out, tot


def assert_grads_close(mine, ref, rtol=3e-3, atol_total=2e-6, median_rtol=5e-4):
    """per parameter: ||g - g_ref|| <= rtol * ||g_ref|| + atol_total * ||all gradients||, and the median over the
    parameters of ||g - g_ref|| / ||g_ref|| <= median_rtol.  Measured (tools/gradcmp.py): median 5e-6 .. 4e-5 with
    the default arithmetic (3e-4 with the native fp32 MFMA), single parameters up to 1.5e-3: two correct fp32
    implementations decide a handful of ReLU masks differently (pre-activations within an ulp of zero, summation orders
    that the autotuner picks per run), and one flipped mask moves a whole row of a weight gradient.  A wiring error --
    transposed taps, a sign, a wrong buffer -- is a relative error of order 1."""
    errs, tot = grad_rel_errors(mine, ref)
    med = float(np.median([d / max(b, 1e-30) for d, b in errs.values()]))
    assert med <= median_rtol, med
    bad = [(n, d, b) for n, (d, b) in errs.items() if d > rtol * b + atol_total * tot]
    worst = max(errs.items(), key=lambda kv: kv[1][0] / max(kv[1][1], 1e-30))
    assert not bad, (bad[:8], "worst", worst)
    return worst


def assert_sampled_grads(named_grads, g, rtol_vec=1e-3, rtol_elem=5e-3, atol_total=0.0, total=None):
    """g = tests/golden/model_grads.npz (sampled gradient elements written by the reference): per parameter the sampled
    vector within rtol_vec of the reference in norm, every element within rtol_elem of the parameter's gradient rms."""
    names = [str(n) for n in g["names"]]
    off, idx, val, rms = g["offsets"], g["idx"], g["val"], g["rms"]
    bad = []
    for k, n in enumerate(names):
        i, v = idx[off[k]:off[k + 1]], val[off[k]:off[k + 1]].astype(np.float64)
        a = named_grads[n].detach().double().cpu().reshape(-1).numpy()[i]
        floor = atol_total * (total or 0.0)
        if np.linalg.norm(a - v) > rtol_vec * max(np.linalg.norm(v), rms[k] * np.sqrt(len(i))) + floor or \
                np.abs(a - v).max() > rtol_elem * rms[k] + floor:
            bad.append((n, float(np.linalg.norm(a - v)), float(np.linalg.norm(v)), float(np.abs(a - v).max()), float(rms[k])))
    assert not bad, bad[:8]
