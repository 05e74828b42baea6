# scratch: k_bloom3 first run -- parity families, then the c3 line, then phases
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -15 > gpurun_out/r4b/tests1.log; tail -6 gpurun_out/r4b/tests1.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-boundary > gpurun_out/r4b/bench.json 2> gpurun_out/r4b/bench.log; echo bench rc=$?
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4b/bench.json'))
print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], 'verified', d.get('verified'), d['build_id'])
PY
BFCG_ABLATE=64 BR=5500000 NB=9 timeout 600 python scripts/bloom_phases.py > gpurun_out/r4b/phases.txt 2>&1
tail -9 gpurun_out/r4b/phases.txt
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py -q -m gpu -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -15 > gpurun_out/r4b/tests2.log; tail -6 gpurun_out/r4b/tests2.log
