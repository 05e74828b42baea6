cd $GRAFT_REPO_ROOT
for r in 7 9; do
BFCG_R=$r timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>gpurun_out/try_r$r.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('R=$r', d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d.get('verified'), d['config']['library_batches_per_step'], d['config']['slow_buckets'])" || tail -3 gpurun_out/try_r$r.log
done
