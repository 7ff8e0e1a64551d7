"""Host budget of a data-parallel rank: bench.py's train step with this process pinned to N cores BEFORE anything touches
the GPU (8 ranks on the GPU box's 16-core cgroup quota leave 2 cores per rank; torch / HIP / RCCL helper threads inherit the
mask).   python tools/bench_pinned.py 2 [bench.py arguments]   -> bench.py's JSON line (host_enqueue_ms_per_step next to
ms_per_step)."""
import os
import runpy
import sys

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cores = sorted(os.sched_getaffinity(0))[:n]
os.sched_setaffinity(0, cores)
os.environ["OMP_NUM_THREADS"] = str(n)
sys.argv = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")] + sys.argv[2:]
print(f"[bench_pinned] affinity {cores}", file=sys.stderr, flush=True)
runpy.run_path(sys.argv[0], run_name="__main__")
