cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/exp5
export TMPDIR=/tmp
O=gpurun_out/exp5
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
run base A=1
run chunk16 BFCG_S1_CHUNK=16
run chunk8 BFCG_S1_CHUNK=8
run chunk64 BFCG_S1_CHUNK=64
