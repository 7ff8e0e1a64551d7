run() { name=$1; shift; env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], d['config']['losses'], d['roofline']['achieved'])
except Exception as e: print('$name FAILED', e)"; }
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv_fwd_dgrad_wgrad or wgrad_group" 2>&1 | tail -2
RADET_WGRAD_BP32=1 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv_fwd_dgrad_wgrad" 2>&1 | tail -2
run base A=1
run bp32 RADET_WGRAD_BP32=1
run base2 A=1
run bp32b RADET_WGRAD_BP32=1
RADET_WGRAD_BP32=1 python tools/prof_layers.py wgrad 2>&1 | tail -42
