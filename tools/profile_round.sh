#!/bin/bash
# Refresh the rocprofv3 evidence kept under profiles/ (run on the GPU box through gpurun):
#   bash tools/profile_round.sh <round tag>        # e.g. round3 -> gpurun_out/<tag>/...
# Kernel traces (stats / timeline) and PMC passes are separate runs; the program is started directly after `--`.
set -u
TAG=${1:-round3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH_SHORT="--no-cpu-baseline --no-mfma-line --no-extras --no-kernel-events"
prof() {   # name, program args...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace -d /tmp/prof_$name -o p -- python3 "$@" > $OUT/${name}_run.log 2>&1
  local db=$(find /tmp/prof_$name -name "*.db" | head -1)
  python3 $R/tools/rocpd_stats.py $db $OUT/${name}_kernel_stats.csv > /dev/null
  echo $db
}
pmc() {    # name, counters (space separated), program args...
  local name=$1; local ctr=$2; shift; shift
  rm -rf /tmp/pmc_${name}
  rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_${name} -o p --output-format csv -- python3 "$@" > $OUT/pmc_${name}.log 2>&1
}
# 1. headline bench: the JSON line (incl. the per-kernel HIP-event table, infer / r101 / trained-like-weights lines), then
#    kernel stats + timeline (phases, idle gaps) of the same step under rocprofv3
python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
db=$(prof bench $R/bench.py --steps 20 --warmup 3 $BENCH_SHORT)
python3 $R/tools/trace_timeline.py $db 5 $OUT/bench_timeline.txt > /dev/null
# 1b. the same step without plane operands for the towers (RADET_P3=0: every GEMM splits in registers), for comparison
RADET_P3=0 python3 $R/bench.py --steps 20 --warmup 5 $BENCH_SHORT > $OUT/bench_p3_off.json 2>> $OUT/bench_default.err
# 2. MFMA utilisation per kernel (SQ counters, one pass)
pmc mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES" $R/bench.py --steps 2 --warmup 1 $BENCH_SHORT
python3 $R/tools/pmc_mfma.py /tmp/pmc_mfma/p_counter_collection.csv $OUT/pmc_mfma.txt > /dev/null
# 3. HBM traffic (FETCH_SIZE / WRITE_SIZE: separate passes) of the dominant kernel of bench.py's `roofline` and of the tower GEMM
for c in FETCH_SIZE WRITE_SIZE; do
  pmc fp32_$c $c $R/bench.py --steps 2 --warmup 1 $BENCH_SHORT
done
DOM=$(python3 -c "import json; d=json.loads([l for l in open('$OUT/bench_default.json') if l.startswith('{')][-1]); print(d['roofline']['kernel']); print(d['roofline']['algorithmic_bytes_per_launch'])")
DOM_K=$(echo "$DOM" | head -1); DOM_B=$(echo "$DOM" | tail -1)
python3 $R/tools/pmc_traffic.py /tmp/pmc_fp32_FETCH_SIZE /tmp/pmc_fp32_WRITE_SIZE "$DOM_K" \
    $OUT/pmc_hbm_traffic_fp32.txt $OUT/roofline_traffic.json $DOM_B > /dev/null 2>> $OUT/bench_default.err
TOW=$(python3 -c "import json; d=json.loads([l for l in open('$OUT/bench_default.json') if l.startswith('{')][-1]); t=d['roofline_tower_forward']; print(t['kernel']); print(t['algorithmic_bytes_per_launch'])")
TOW_K=$(echo "$TOW" | head -1); TOW_B=$(echo "$TOW" | tail -1)
python3 $R/tools/pmc_traffic.py /tmp/pmc_fp32_FETCH_SIZE /tmp/pmc_fp32_WRITE_SIZE "$TOW_K" \
    $OUT/pmc_hbm_traffic_tower.txt $OUT/roofline_traffic_tower.json $TOW_B > /dev/null 2>> $OUT/bench_default.err
# 4. inference (config 4) kernel stats, R101 (config 5) log, fill-path micro-benchmark
db=$(prof infer $R/tools/bench_configs.py infer --images 400)
python3 $R/tools/trace_timeline.py --infer $db 7 $OUT/infer_timeline.txt > /dev/null
python3 $R/tools/bench_configs.py infer > $OUT/infer.log 2>&1
python3 $R/tools/bench_latency.py > $OUT/latency_b1.log 2>&1
export LAT_NO_GRAPH=1
db=$(prof latency $R/tools/bench_latency.py)
unset LAT_NO_GRAPH
python3 $R/tools/trace_timeline.py --infer $db 21 $OUT/latency_b1_timeline.txt > /dev/null
python3 $R/tools/bench_configs.py r101 > $OUT/r101.log 2>&1
[ -x $R/tools/_probe/fill_probe ] && $R/tools/_probe/fill_probe > $OUT/fill_probe.txt 2>&1
ls -la $OUT
