cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py 2>gpurun_out/default_bench.log > gpurun_out/default_bench.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/default_bench.json'))
print({k: d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','higher_is_better','scaling','vs_baseline','dtype','data','verified','build_id')})
print(d['roofline']['frac'], d['roofline'].get('traffic_frac'), d['roofline'].get('traffic_note','')[-40:], d['cpu_baseline']['value'], d['cpu_baseline']['kind'])
PY
