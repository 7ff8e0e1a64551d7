#!/bin/bash
# Regenerate the tune file on this box and compare it with the shipped one (run through gpurun):
#   bash tools/compare_tune.sh        -> gpurun_out/tune_new.json + one bench line per (file, arithmetic mode)
# The shipped radet_amd/tune_gfx950.json is never touched: the candidate is selected through RADET_TUNE_FILE (its entries
# override the packaged ones key by key); the "shipped" leg points RADET_TUNE_FILE at an empty scratch file so that no user
# cache of this machine leaks into either measurement.
mkdir -p gpurun_out
python tools/make_tune.py gpurun_out/tune_new.json > gpurun_out/tune_new.log 2>&1
EMPTY=$(mktemp /tmp/tune_empty.XXXXXX.json)
trap 'rm -f "$EMPTY"' EXIT
echo '{"igemm": {}, "wgrad": {}}' > "$EMPTY"
for f in "$EMPTY" gpurun_out/tune_new.json; do
  for m in fp32 bf16-storage; do
    cp "$f" /tmp/tune_leg.json          # (unknown shapes tuned during the run are appended to the copy, not to the candidate)
    RADET_TUNE_FILE=/tmp/tune_leg.json python bench.py --math $m --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f', '$m', d['value'], d['ms_per_step'])"
  done
done
