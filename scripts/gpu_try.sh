cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python scripts/c4_run.py --batch-reads 16777216 > gpurun_out/r2b_c4_16m.log 2>&1; echo c4 rc=$?; tail -4 gpurun_out/r2b_c4_16m.log | cut -c1-600
timeout 1500 python scripts/c4_run.py --k 51 --filter-mode 1 --trim 1 --batch-reads 8388608 > gpurun_out/r2b_c5_8m.log 2>&1; echo c5 rc=$?; tail -5 gpurun_out/r2b_c5_8m.log | cut -c1-600
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/try_bench.json 2>gpurun_out/try_bench.log; python - <<'PY'
import json
d=json.load(open('gpurun_out/try_bench.json'))
print(d['value'], d['ms_per_step'], d['config']['partition'], list(d['stages']), d['stages']['level2']['frac'], d['roofline'].get('traffic_note'))
PY
