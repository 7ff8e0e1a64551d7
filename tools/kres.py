#!/usr/bin/env python
"""Per-kernel resource table from `hipcc -Rpass-analysis=kernel-resource-usage` remarks.
usage: tools/kres.py <file.hip> [name regex]   (compiles to /tmp; prints VGPRs / scratch / LDS / occupancy)"""
import re
import subprocess
import sys

src = sys.argv[1]
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Iinclude", "-I../../include",
       "-mllvm", "-amdgpu-mfma-vgpr-form", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for ln in err.splitlines():
    m = re.search(r"remark: \s*(.+?)(?: \[-Rpass)", ln)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem).replace("void ", "")
    if pat and not pat.search(dem):
        continue
    print(f"{dem:70s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>3s} scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} "
          f"lds {r.get('LDS Size [bytes/block]','?'):>6s} occ {r.get('Occupancy [waves/SIMD]','?')}")
