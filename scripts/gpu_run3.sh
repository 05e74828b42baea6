cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu -k "baseline_shapes and not c3_full and not c4_param and not c5_param or fuzz and (random_configuration or medium)" -x 2>&1 | tail -5
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench3.json 2> gpurun_out/r2_bench3.log; echo bench rc=$?
tail -2 gpurun_out/r2_bench3.log; python - <<'PY'
import json
d=json.load(open('gpurun_out/r2_bench3.json'))
print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d.get('verified'))
print(d['secondary']['c2']['value'], d['secondary']['c2']['stage_ms_per_step'], d['secondary']['c2'].get('verified'))
PY
PMC=2 STEPS=1 bash scripts/prof_round2.sh r2c 2>&1 | tail -3
python tools/make_round2_md.py gpurun_out/prof_r2c c3 > gpurun_out/round2_c3.md; cp profiles/round2_c3_pmc.json gpurun_out/; tail -30 gpurun_out/round2_c3.md
