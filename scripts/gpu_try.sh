cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_group.py tests/test_gpu_dropin.py -q -m gpu -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4
timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x -k "emulated" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
BFC_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --batch-reads 2097152 2>gpurun_out/try_dist.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('world-1 RCCL path:', d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['config']['partition'], d.get('verified'))"
