#!/bin/bash
# Refresh the rocprofv3 evidence kept under profiles/ (run on the GPU box through gpurun):
#   bash tools/profile_round.sh <round tag>        # e.g. round2 -> gpurun_out/<tag>/...
# Kernel traces (stats / timeline) and PMC passes are separate runs; the program is started directly after `--`.
set -u
TAG=${1:-round2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() {   # name, program args...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace -d /tmp/prof_$name -o p -- python3 "$@" > $OUT/${name}_run.log 2>&1
  local db=$(find /tmp/prof_$name -name "*.db" | head -1)
  python3 $R/tools/rocpd_stats.py $db $OUT/${name}_kernel_stats.csv > /dev/null
  echo $db
}
pmc() {    # name, counter, program args...
  local name=$1; local ctr=$2; shift; shift
  rm -rf /tmp/pmc_${name}_$ctr
  rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_${name}_$ctr -o p --output-format csv -- python3 "$@" > $OUT/pmc_${name}_$ctr.log 2>&1
}
# 1. headline bench: JSON line, kernel stats, timeline
python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
db=$(prof bench $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-mfma-line)
python3 $R/tools/trace_timeline.py $db 5 $OUT/bench_timeline.txt > /dev/null
# 1b. the same step with the native fp32 matrix instruction
python3 $R/bench.py --math fp32-mfma --no-cpu-baseline > $OUT/bench_fp32_mfma.json 2>> $OUT/bench_default.err
db=$(prof bench_fp32_mfma $R/bench.py --math fp32-mfma --steps 20 --warmup 3 --no-cpu-baseline)
# 2. bf16-storage (BASELINE config 3 arithmetic)
python3 $R/bench.py --math bf16-storage --no-cpu-baseline > $OUT/bench_bf16_storage.json 2>> $OUT/bench_default.err
db=$(prof bench_bf16_storage $R/bench.py --math bf16-storage --steps 20 --warmup 3 --no-cpu-baseline)
# 3. inference (config 4) and R101 800x800 (config 5)
python3 $R/tools/bench_configs.py infer > $OUT/infer.log 2>&1
db=$(prof infer $R/tools/bench_configs.py infer)
python3 $R/tools/bench_configs.py r101 > $OUT/r101.log 2>&1
# 4. PMC: HBM traffic of the fp32 and bf16-storage GEMM kernels (separate passes per counter)
for c in FETCH_SIZE WRITE_SIZE; do
  pmc fp32 $c $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-mfma-line
  pmc fp32m $c $R/bench.py --math fp32-mfma --steps 2 --warmup 1 --no-cpu-baseline
  pmc bf16s $c $R/bench.py --math bf16-storage --steps 2 --warmup 1 --no-cpu-baseline
done
python3 $R/tools/pmc_traffic.py /tmp/pmc_fp32_FETCH_SIZE /tmp/pmc_fp32_WRITE_SIZE "conv_igemmg_kernel<128, 128, 2, 2, 9, 32, 2, false>" \
    $OUT/pmc_hbm_traffic_fp32.txt $OUT/roofline_traffic.json 103022592 > /dev/null 2>> $OUT/bench_default.err
python3 $R/tools/pmc_traffic.py /tmp/pmc_fp32m_FETCH_SIZE /tmp/pmc_fp32m_WRITE_SIZE "conv_igemmg_kernel<128, 64, 2, 2, 1, 32, 3, false>" \
    $OUT/pmc_hbm_traffic_fp32_mfma.txt $OUT/roofline_traffic_fp32_mfma.json 103022592 > /dev/null 2>> $OUT/bench_default.err
python3 $R/tools/pmc_traffic.py /tmp/pmc_bf16s_FETCH_SIZE /tmp/pmc_bf16s_WRITE_SIZE "conv_igemmg_kernel<128, 64, 2, 2, 5, 32, 2, false>" \
    $OUT/pmc_hbm_traffic_bf16_storage.txt $OUT/roofline_traffic_bf16_storage.json 51511296 > /dev/null 2>> $OUT/bench_default.err
ls -la $OUT
