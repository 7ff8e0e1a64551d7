S="--no-cpu-baseline --no-mfma-line --no-extras --no-kernel-events --steps 40"
for n in 16 2 1; do python tools/bench_pinned.py $n $S 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cores $n', d['ms_per_step'], 'ms/step, host enqueue', d['host_enqueue_ms_per_step'])"; done
python tools/bench_user_stream.py 40
python -m pytest tests/test_gpu_configs.py tests/test_gpu_model.py -x -q -s -k "r101_800 or streamed_inference or stream_budget or abandoned or rccl" 2>&1 | tail -15
