# round 3: full GPU suite + smoke + a driver-style bench line
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > gpurun_out/r3_gpu_tests.log; tail -8 gpurun_out/r3_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps ${STEPS:-10} --warmup 3 > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.log; echo bench rc=$?
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_bench.json'))
print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], 'verified', d.get('verified'), d['build_id'])
print(d.get('secondary',{}).get('c2',{}).get('value'), d.get('secondary',{}).get('c2',{}).get('verified'), (d.get('cpu_baseline') or {}).get('value'))
PY
