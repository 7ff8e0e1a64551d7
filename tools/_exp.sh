run() { name=$1; shift; python bench.py "$@" --steps 30 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], d['config']['losses'], d.get('roofline',{}).get('achieved'))
except Exception as e: print('$name FAILED', e)"; }
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv or wgrad" 2>&1 | tail -2
run fp32
run fp32b
run bf16s --math bf16-storage
python tools/prof_layers.py 2>&1 | tail -1
