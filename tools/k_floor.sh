#!/bin/bash
# kernel durations (rocprofv3 timestamps, not launch rate) of the implicit-GEMM kernel vs K:  bash tools/k_floor.sh H W N [tile]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kf
rocprofv3 --kernel-trace -d /tmp/kf -o p -- python3 $R/tools/k_scaling.py "$@" > /tmp/kf.log 2>&1
db=$(find /tmp/kf -name "*.db" | head -1)
python3 - $db <<'P'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, duration, start from kernels order by start").fetchall()
rows = [(n, d, s) for n, d, s in rows if "conv_igemmg" in n]
# k_scaling runs 7 K values x (5 warm + 100 timed) launches in order
per = len(rows) // 7
for i, K in enumerate((32, 64, 128, 256, 512, 1024, 2048)):
    chunk = rows[i * per:(i + 1) * per][5:]
    d = sorted(r[1] for r in chunk)
    gaps = sorted(chunk[j + 1][2] - (chunk[j][2] + chunk[j][1]) for j in range(len(chunk) - 1))
    print(f"K={K:5d}: kernel duration median {d[len(d) // 2] / 1e3:6.1f} us  min {d[0] / 1e3:6.1f}   gap to next launch median {gaps[len(gaps) // 2] / 1e3:5.1f} us")
P
