"""Timeline view of a rocprofv3 (rocpd sqlite) kernel trace of bench.py: where one steady-state train step spends its
time.  Steps are delimited by `adamw_kernel`; for the median of the last N steps it prints the wall time, the time
the GPU ran >= 1 kernel (union), the summed kernel time, per-queue busy time, the largest idle gaps (with the kernels
around them) and a per-kernel table (sum / calls / share of the union).

Usage: python tools/trace_timeline.py <results.db> [n_steps=5] [out.txt]"""
import collections
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name).replace("void ", "")
    return name[:70]


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    gaps = []
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            gaps.append((s - ce, ce, s))
            cs, ce = s, e
    if cs is not None:
        tot += ce - cs
    return tot, gaps


def infer_main():
    """python tools/trace_timeline.py --infer <results.db> [n_batches=5] [out.txt]: one steady-state batch of the streamed
    inference loop (rt.detect_stream): batches are delimited by the start of `stem_kernel`; decode + NMS of batch k run
    on another queue next to the forward pass of batch k + 1, so a batch window [stem k, stem k + 1) holds the forward
    pass of batch k and the post-processing of batch k - 1."""
    db = sqlite3.connect(sys.argv[2])
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    rows = db.execute("select name, start, end, queue_id from kernels order by start").fetchall()
    rows = [(short(a), s, e, q) for a, s, e, q in rows]
    starts = [s for a, s, e, q in rows if a.startswith("stem_kernel")]
    if len(starts) < n + 2:
        raise SystemExit(f"only {len(starts)} batches in the trace")
    wins = []
    for i in range(len(starts) - n - 1, len(starts) - 1):
        t0, t1 = starts[i], starts[i + 1]
        ks = [r for r in rows if r[1] >= t0 and r[1] < t1]
        busy, gaps = union([(s, min(e, t1)) for _, s, e, _ in ks])
        wins.append((t1 - t0, busy, sum(e - s for _, s, e, _ in ks), ks, gaps, t0, t1))
    wins.sort(key=lambda x: x[0])
    wall, busy, ksum, ks, gaps, t0, t1 = wins[len(wins) // 2]
    post = ("decode_kernel", "compact_levels", "nms_", "at::native", "rocprim", "__amd_rocclr")
    fwd = [r for r in ks if not r[0].startswith(post)]
    pp = [r for r in ks if r[0].startswith(post)]
    out = [f"median of last {n} batch windows (stem k .. stem k+1): wall {wall / 1e6:.3f} ms | GPU busy (union) {busy / 1e6:.3f} ms | idle "
           f"{(wall - busy) / 1e6:.3f} ms | summed kernel time {ksum / 1e6:.3f} ms (overlap x{ksum / busy:.2f}) | {len(ks)} launches",
           "all windows wall ms: " + " ".join(f"{w[0] / 1e6:.2f}" for w in wins)]
    perq = collections.defaultdict(int)
    for _, s, e, q in ks:
        perq[q] += e - s
    out.append("per-queue busy ms: " + ", ".join(f"q{q}: {v / 1e6:.2f}" for q, v in sorted(perq.items())))
    gaps.sort(reverse=True)
    out.append(f"idle gaps: {len(gaps)}; > 5 us: {sum(1 for g in gaps if g[0] > 5000)} totalling {sum(g[0] for g in gaps if g[0] > 5000) / 1e6:.3f} ms; top 8:")
    for g, a, b in gaps[:8]:
        before = [r[0] for r in ks if r[2] == a][:1]
        after = [r[0] for r in ks if r[1] == b][:1]
        out.append(f"   {g / 1e3:8.1f} us at +{(a - t0) / 1e6:6.3f} ms   after {before}  before {after}")
    def first(prefix, seq=ks):
        c = [r for r in seq if r[0].startswith(prefix)]
        return c[0] if c else None
    def last_end(prefixes, seq=ks):
        c = [r[2] for r in seq if r[0].startswith(prefixes)]
        return max(c) if c else None
    marks = [("backbone + neck forward", t0), ("head forward (towers + predictors)", (first("split_planes_kernel") or first("gn_stats_kernel") or (0, t0))[1])]
    head_end = last_end(("pred3x3_patch_kernel", "conv_igemmg_kernel", "gn_apply_kernel"), fwd)
    out.append("phases of the forward pass of batch k (wall):")
    out.append(f"   backbone + neck forward            +  0.000 ms  {(marks[1][1] - t0) / 1e6:7.3f} ms   "
               f"summed kernel time {sum(e - s for a, s, e, q in fwd if s < marks[1][1]) / 1e6:7.3f} ms in {sum(1 for r in fwd if r[1] < marks[1][1])} launches")
    out.append(f"   head forward (towers, predictors)  +{(marks[1][1] - t0) / 1e6:7.3f} ms  {(head_end - marks[1][1]) / 1e6:7.3f} ms   "
               f"summed kernel time {sum(e - s for a, s, e, q in fwd if s >= marks[1][1]) / 1e6:7.3f} ms in {sum(1 for r in fwd if r[1] >= marks[1][1])} launches")
    if pp:
        p0, p1 = min(r[1] for r in pp), max(r[2] for r in pp)
        out.append(f"   decode + NMS of batch k - 1        +{(p0 - t0) / 1e6:7.3f} ms  {(p1 - p0) / 1e6:7.3f} ms   summed kernel time "
                   f"{sum(e - s for _, s, e, _ in pp) / 1e6:7.3f} ms in {len(pp)} launches (other queue, next to the forward pass)")
    agg = collections.defaultdict(lambda: [0, 0])
    for a, s, e, q in ks:
        agg[a][0] += 1
        agg[a][1] += e - s
    out.append(f"{'kernel':72s} calls   sum_ms  avg_us  %busy")
    for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
        out.append(f"{k:72s} {c:5d} {d / 1e6:8.3f} {d / c / 1e3:7.1f} {100.0 * d / busy:6.1f}")
    txt = "\n".join(out)
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write(txt + "\n")
    print(txt)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--infer":
        return infer_main()
    db = sqlite3.connect(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    rows = db.execute("select name, start, end, queue_id from kernels order by start").fetchall()
    rows = [(short(a), s, e, q) for a, s, e, q in rows]
    ends = [e for a, s, e, q in rows if a.startswith("adamw_kernel")]
    if len(ends) < n + 1:
        raise SystemExit(f"only {len(ends)} steps in the trace")
    out = []
    steps = []
    for i in range(len(ends) - n, len(ends)):
        t0, t1 = ends[i - 1], ends[i]
        ks = [r for r in rows if r[1] >= t0 and r[2] <= t1]
        busy, gaps = union([(s, e) for _, s, e, _ in ks])
        steps.append((t1 - t0, busy, sum(e - s for _, s, e, _ in ks), ks, gaps, t0))
    steps.sort(key=lambda x: x[0])
    wall, busy, ksum, ks, gaps, t0 = steps[len(steps) // 2]
    out.append(f"median of last {n} steps: wall {wall / 1e6:.3f} ms | GPU busy (union) {busy / 1e6:.3f} ms | idle {(wall - busy) / 1e6:.3f} ms | "
               f"summed kernel time {ksum / 1e6:.3f} ms (overlap x{ksum / busy:.2f}) | {len(ks)} launches")
    out.append("all steps wall ms: " + " ".join(f"{s[0] / 1e6:.2f}" for s in steps))
    perq = collections.defaultdict(int)
    for _, s, e, q in ks:
        perq[q] += e - s
    out.append("per-queue busy ms: " + ", ".join(f"q{q}: {v / 1e6:.2f}" for q, v in sorted(perq.items())))
    gaps.sort(reverse=True)
    out.append(f"idle gaps: {len(gaps)}; > 5 us: {sum(1 for g in gaps if g[0] > 5000)} totalling {sum(g[0] for g in gaps if g[0] > 5000) / 1e6:.3f} ms; top 12:")
    for g, a, b in gaps[:12]:
        before = [r[0] for r in ks if r[2] == a][:1]
        after = [r[0] for r in ks if r[1] == b][:1]
        out.append(f"   {g / 1e3:8.1f} us at +{(a - t0) / 1e6:6.3f} ms   after {before}  before {after}")
    agg = collections.defaultdict(lambda: [0, 0])
    for a, s, e, q in ks:
        agg[a][0] += 1
        agg[a][1] += e - s
    out.append(f"{'kernel':72s} calls   sum_ms  avg_us  %busy")
    for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append(f"{k:72s} {c:5d} {d / 1e6:8.3f} {d / c / 1e3:7.1f} {100.0 * d / busy:6.1f}")
    # phases of the step, delimited by marker kernels (start of the first occurrence)
    def first(prefix, last=False):
        c = [r for r in ks if r[0].startswith(prefix)]
        return (c[-1] if last else c[0]) if c else None
    marks = [("backbone + neck forward", first("stem_kernel")), ("head forward", first("gn_stats_kernel")),
             ("loss", first("loss_prep_rows_kernel")), ("head backward", first("gn_bwd_stats_kernel")),
             ("neck backward", first("upsample_add_bwd_kernel")), ("backbone backward", first("relu_bwd_kernel")),
             ("clip + AdamW", first("sqnorm_kernel"))]
    marks = [(n, r[1]) for n, r in marks if r is not None]
    if marks:
        out.append("phases (wall between marker kernels; the head-forward marker is the first GroupNorm, one tower GEMM late):")
        for i, (n, t) in enumerate(marks):
            t_next = marks[i + 1][1] if i + 1 < len(marks) else t0 + wall
            seg = [(a, st, e) for a, st, e, _ in ks if st >= t and st < t_next]
            out.append(f"   {n:28s} +{(t - t0) / 1e6:7.3f} ms  {(t_next - t) / 1e6:7.3f} ms   summed kernel time "
                       f"{sum(e - st for _, st, e in seg) / 1e6:7.3f} ms in {len(seg)} launches")
    txt = "\n".join(out)
    if len(sys.argv) > 4:                  # flat kernel list of the median step
        with open(sys.argv[4], "w") as fh:
            fh.write("name,start_us,dur_us,queue\n")
            for a, st, e, q in ks:
                fh.write(f'"{a}",{(st - t0) / 1e3:.1f},{(e - st) / 1e3:.1f},{q}\n')
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
