#!/bin/bash
# rocprofv3 evidence for the bf16-storage step (BASELINE configs[2] arithmetic on one GPU), run on the GPU box through gpurun:
#   bash tools/profile_bf16.sh <round tag>      -> gpurun_out/<tag>_bf16/...
# Kernel trace and PMC passes are separate runs; the program is started directly after `--`.
set -u
TAG=${1:-round6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_bf16
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHORT="--math bf16-storage --no-cpu-baseline --no-mfma-line --no-extras --no-clock-sampler"
python3 $R/bench.py $SHORT > $OUT/bench_bf16_storage.json 2> $OUT/bench.err
rm -rf /tmp/prof_b16
rocprofv3 --kernel-trace -d /tmp/prof_b16 -o p -- python3 $R/bench.py --steps 20 --warmup 3 $SHORT --no-kernel-events > $OUT/prof_run.log 2>&1
db=$(find /tmp/prof_b16 -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $db $OUT/bf16_storage_kernel_stats.csv > /dev/null
python3 $R/tools/trace_timeline.py $db 5 $OUT/bf16_storage_timeline.txt > /dev/null
pmc() {
  local name=$1; local ctr=$2; shift; shift
  rm -rf /tmp/pmc_${name}
  rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_${name} -o p --output-format csv -- python3 "$@" > $OUT/pmc_${name}.log 2>&1
}
pmc b16mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES" $R/bench.py --steps 2 --warmup 1 $SHORT --no-kernel-events
python3 $R/tools/pmc_mfma.py /tmp/pmc_b16mfma/p_counter_collection.csv $OUT/bf16_storage_pmc_mfma.txt > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  pmc b16_$c $c $R/bench.py --steps 2 --warmup 1 $SHORT --no-kernel-events
done
DOM=$(python3 -c "import json; d=json.loads([l for l in open('$OUT/bench_bf16_storage.json') if l.startswith('{')][-1]); print(d['roofline']['kernel']); print(d['roofline'].get('algorithmic_bytes_per_launch', 0))")
DOM_K=$(echo "$DOM" | head -1); DOM_B=$(echo "$DOM" | tail -1)
python3 $R/tools/pmc_traffic.py /tmp/pmc_b16_FETCH_SIZE /tmp/pmc_b16_WRITE_SIZE "$DOM_K" \
    $OUT/pmc_hbm_traffic_bf16_storage.txt $OUT/roofline_traffic_bf16_storage.json $DOM_B > /dev/null 2>> $OUT/bench.err
ls -la $OUT
