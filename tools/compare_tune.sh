#!/bin/bash
# Regenerate the tune file on this box and compare it with the shipped one (run through gpurun):
#   bash tools/compare_tune.sh        -> gpurun_out/tune_new.json + one bench line per (file, arithmetic mode)
mkdir -p gpurun_out
python tools/make_tune.py gpurun_out/tune_new.json > gpurun_out/tune_new.log 2>&1
cp radet_amd/tune_gfx950.json /tmp/old.json
for f in /tmp/old.json gpurun_out/tune_new.json; do
  cp $f radet_amd/tune_gfx950.json
  for m in fp32 bf16-storage; do
    python bench.py --math $m --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f', '$m', d['value'], d['ms_per_step'])"
  done
done
