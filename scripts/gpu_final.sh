# final evidence of the round: full GPU suite, driver-style bench, rocprofv3 traces + PMC passes of c3 and c2
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.log; echo bench rc=$?
PMC=2 STEPS=1 bash scripts/prof_round2.sh c3 > gpurun_out/prof_c3.out 2>&1; tail -3 gpurun_out/prof_c3.out | cut -c1-200
PMC=2 STEPS=3 BENCH_ARGS="--workload c2" bash scripts/prof_round2.sh c2 > gpurun_out/prof_c2.out 2>&1; tail -3 gpurun_out/prof_c2.out | cut -c1-200
python tools/make_round_md.py gpurun_out/prof_c3 c3 > gpurun_out/round2_c3.md; cp profiles/round2_c3_pmc.json gpurun_out/
python tools/make_round_md.py gpurun_out/prof_c2 c2 > gpurun_out/round2_c2.md; cp profiles/round2_c2_pmc.json gpurun_out/
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2_bench_final.json'))
print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], 'verified', d.get('verified'), d['build_id'])
print(d['secondary']['c2']['value'], d['secondary']['c2'].get('verified'), d['cpu_baseline']['value'])
PY
timeout 900 python scripts/gz_rate.py 50 6 4 2>&1 | tee gpurun_out/round2_gz_rate.txt | tail -8
