# round 4 experiments (one gpurun call): the commit probe, then c3 A/Bs by environment switch on ONE box
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/exp1
export TMPDIR=/tmp
O=gpurun_out/exp1
timeout 300 build/commit_probe 17 13 1335 2 > $O/commit_probe_2p.txt 2>&1; tail -14 $O/commit_probe_2p.txt | cut -c1-200
timeout 300 build/commit_probe 17 13 1335 1 > $O/commit_probe_1p.txt 2>&1; tail -7 $O/commit_probe_1p.txt | cut -c1-200
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
run nofast2 BFCG_NO_FAST_S2=1
run pipe BFCG_PIPELINE=1
run pipe_wg1 BFCG_PIPELINE=1 BFCG_S1_WGS=1
run base2 A=1
