cd $GRAFT_REPO_ROOT
BFC_FUZZ_SEED_BASE=21 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x -k "emulated" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -1
timeout 600 python -m pytest tests/test_gpu_group.py tests/test_gpu_fuzz.py -q -m gpu -x -k "group or emulated" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.log; echo bench rc=$?
PMC=1 STEPS=1 bash scripts/prof_round2.sh c3 > gpurun_out/prof_c3.out 2>&1
PMC=1 STEPS=3 BENCH_ARGS="--workload c2" bash scripts/prof_round2.sh c2 > gpurun_out/prof_c2.out 2>&1
python tools/make_round2_md.py gpurun_out/prof_c3 c3 > gpurun_out/round2_c3.md; cp profiles/round2_c3_pmc.json gpurun_out/
python tools/make_round2_md.py gpurun_out/prof_c2 c2 > gpurun_out/round2_c2.md; cp profiles/round2_c2_pmc.json gpurun_out/
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2_bench_final.json'))
print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], 'verified', d.get('verified'), d['build_id'])
print(d['secondary']['c2']['value'], d['secondary']['c2'].get('verified'))
print(json.load(open('gpurun_out/round2_c3_pmc.json'))['build_id'])
PY
