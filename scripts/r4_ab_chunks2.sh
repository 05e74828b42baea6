cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/exp6
export TMPDIR=/tmp
O=gpurun_out/exp6
timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -x -m gpu -k "bit_planes" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_dropin.py -q -x -m gpu 2>&1 | tail -3
bash scripts/e2e_c3.sh > $O/e2e_c3.txt 2>&1; tail -22 $O/e2e_c3.txt | cut -c1-300
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-boundary > $O/$name.json 2> $O/$name.log
  python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], d['ms_per_step'], d['config']['stage_ms_per_step'], 'verified', d.get('verified'), 'frac', d['roofline']['frac'], d['roofline'].get('whole_job_frac'))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run c32a A=1
run c64a BFCG_S1_CHUNK=64
run c128a BFCG_S1_CHUNK=128
run c32b A=1
run c64b BFCG_S1_CHUNK=64
run c128b BFCG_S1_CHUNK=128
