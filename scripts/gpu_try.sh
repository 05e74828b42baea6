cd $GRAFT_REPO_ROOT
for v in 0 1; do
BFCG_ONEPASS=$v BFC_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-verify 2>gpurun_out/try_dist$v.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('ONEPASS=$v world-1 RCCL path:', d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['config']['library_batches_per_step'], d['config']['exchange_plus_stages_s_per_step'])"
done
tail -5 gpurun_out/try_dist1.log
