S="--no-cpu-baseline --no-mfma-line --no-extras --no-kernel-events --steps 40"
for i in 1 2; do
for v in 0 1; do RADET_TOWER_PATCH=$v python bench.py $S 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('patch=$v', d['ms_per_step'], 'ms', d['value'], 'img/s', d['config']['losses_step1'], d['config']['losses'])"; done; done
RADET_TOWER_PATCH=1 python -m pytest tests/test_gpu_model.py -x -q -k "train_forward_backward or simple_test or features" 2>&1 | tail -3
