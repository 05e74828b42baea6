cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_baseline_shapes.py -q -m gpu -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/r2_bench4.json 2> gpurun_out/r2_bench4.log; echo bench rc=$?
tail -2 gpurun_out/r2_bench4.log; python - <<'PY'
import json
d=json.load(open('gpurun_out/r2_bench4.json'))
print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], 'verified', d.get('verified'))
print(d.get('verification'))
print(d['roofline'])
print(d['roofline_bloom'])
print(d['secondary']['c2']['value'], d['secondary']['c2'].get('verified'))
print(d.get('cpu_baseline'))
PY
