# scratch: warm-batch config of k_bloom3 + flattened k_commit_seg: A/B on one box, then the whole GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
export TMPDIR=/tmp
run() { name=$1; shift; env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-boundary > gpurun_out/r4e/$name.json 2> gpurun_out/r4e/$name.log; python - $name <<'PY'
import json, sys
d=json.load(open('gpurun_out/r4e/%s.json' % sys.argv[1]))
print(sys.argv[1], d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], 'verified', d.get('verified'), d['config']['slow_buckets'], d['config']['library_batches_per_step'], d['build_id'])
PY
}
run A_default X=1
run B_nowarm BFCG_B3_WARM=0
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > gpurun_out/r4e/gpu_tests.log; tail -8 gpurun_out/r4e/gpu_tests.log
