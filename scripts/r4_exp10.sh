cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/exp10
export TMPDIR=/tmp
O=gpurun_out/exp10
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x -m gpu 2>&1 | tail -3
for w in c3 c2; do
timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-boundary > $O/$w.json 2> $O/$w.log
python -c "
import json; d=json.load(open('$O/$w.json')); print('$w', d['ms_per_step'], d['config']['stage_ms_per_step'], 'verified', d.get('verified'), 'frac', d['roofline']['frac'], d['roofline'].get('whole_job_frac'))"
done
