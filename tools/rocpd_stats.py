"""Summarise a rocprofv3 (rocpd sqlite) kernel trace like `--stats`: per-kernel calls / total / avg / %.
Usage: python tools/rocpd_stats.py gpurun_out/prof/bench_results.db [out.csv]"""
import collections
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, duration from kernels").fetchall()
    agg = collections.defaultdict(lambda: [0, 0, 1 << 62, 0])
    for n, d in rows:
        a = agg[short(n)]
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f'"{k}",{a[0]},{a[1]},{a[1] / a[0]:.0f},{100.0 * a[1] / total:.2f},{a[2]},{a[3]}')
    out = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")
    print(out)


if __name__ == "__main__":
    main()
