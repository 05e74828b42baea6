cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_baseline_shapes.py tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/try_bench.json 2> gpurun_out/try_bench.log; echo rc=$?
python - <<'PY'
import json
d=json.load(open('gpurun_out/try_bench.json'))
print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], 'verified', d.get('verified'))
print(d['secondary']['c2']['value'], d['secondary']['c2'].get('verified'), d['secondary']['c2'].get('stage_ms_per_step'))
PY
BFCG_ONEPASS2=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('two-pass level 2:', d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'])"
