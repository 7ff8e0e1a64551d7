#!/bin/bash
# Shader clock / power sampled while the headline step runs: bash tools/clock_watch.sh <out file> [extra bench.py arguments]
# (rocm-smi polled every ~0.2 s next to `python bench.py --steps 2000 ...`; the summary keeps the samples taken under load)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$(realpath -m ${1:-gpurun_out/clock_watch.txt}); shift
python3 $R/bench.py --steps 2000 --warmup 5 --no-cpu-baseline --no-mfma-line --no-extras --no-kernel-events "$@" > /tmp/cw_bench.json 2>/dev/null &
BP=$!
: > /tmp/cw_raw.txt
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Socket Graphics Package Power" | tr -s ' \t' ' ' | tr '\n' ' ' >> /tmp/cw_raw.txt
  echo >> /tmp/cw_raw.txt
  sleep 0.2
done
wait $BP
python3 - > $OUT <<'PY'
import re
rows = []
for ln in open('/tmp/cw_raw.txt'):
    m = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", ln); p = re.search(r"Power \(W\): ([0-9.]+)", ln)
    if m and p:
        rows.append((int(m.group(1)), float(p.group(1))))
load = [r for r in rows if r[1] > 500]
print(f"{len(rows)} samples, {len(load)} under load (> 500 W)")
if load:
    s = sorted(r[0] for r in load); w = sorted(r[1] for r in load)
    print(f"sclk under load: min {s[0]} median {s[len(s)//2]} max {s[-1]} MHz;  power: min {w[0]:.0f} median {w[len(w)//2]:.0f} max {w[-1]:.0f} W")
    print("samples (MHz, W):", " ".join(f"{a}/{b:.0f}" for a, b in load[:40]))
PY
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' /tmp/cw_bench.json | head -2 >> $OUT
cat $OUT
