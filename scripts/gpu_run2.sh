cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -k "not c3_full and not c4_param and not c5_param" 2>&1 | tail -15
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench2.json 2> gpurun_out/r2_bench2.log; echo bench rc=$?
tail -3 gpurun_out/r2_bench2.log; python - <<'PY'
import json
d=json.load(open('gpurun_out/r2_bench2.json'))
print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d.get('verified'))
print(json.dumps(d.get('secondary'))[:600])
PY
STEPS=1 bash scripts/prof_round2.sh r2b 2>&1 | tail -45
