#!/bin/bash
# Counter passes over one conv launch (tools/pmc_one.py, or PMC_PROG=tools/pmc_wgrad9.py): bash tools/pmc_one.sh <out file> <program arguments...>
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$(realpath -m $1); shift
cd /tmp && export TMPDIR=/tmp
: > $OUT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY" \
           "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ"; do
  rm -rf /tmp/pmc_one
  timeout 150 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_one -o p --output-format csv -- python3 $R/${PMC_PROG:-tools/pmc_one.py} "$@" > /tmp/pmc_one.log 2>&1
  python3 - "$set" >> $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob('/tmp/pmc_one/**/*counter_collection.csv', recursive=True)
if not f:
    print("no counter file for", sys.argv[1], open('/tmp/pmc_one.log').read()[-600:]); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    kn = r['Kernel_Name']
    if 'conv_igemmg' not in kn and 'wgrad' not in kn: continue
    acc[kn][r['Counter_Name']] += float(r['Counter_Value']); n[kn, r['Counter_Name']] += 1
for kn, d in acc.items():
    print(kn[:70], {c: round(v / n[kn, c]) for c, v in d.items()})
PY
done
cat $OUT
