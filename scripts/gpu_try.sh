cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_baseline_shapes.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/try.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d.get('verified')); print(d['secondary']['c2']['value'], d['secondary']['c2'].get('verified'), d['secondary']['c2'].get('stage_ms_per_step'))"
