cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
: > gpurun_out/run24.log
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 >> gpurun_out/run24.log
for i in 1 2; do python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['roofline']['frac'], d['config']['slow_buckets'], d['config']['tab_cshift'])
" >> gpurun_out/run24.log 2>&1; done
BFCG_SYNC_BATCHES=1 timeout 600 python scripts/c3_run.py --b 35 --batch-reads 2097152 --digest 0 --cov 10 2>&1 | grep -v "^\[c3\]" | cut -c1-500 >> gpurun_out/run24.log
timeout 600 python scripts/c3_run.py --b 35 --batch-reads 2097152 --digest 0 2>&1 | grep -v "^\[c3\]" | cut -c1-500 >> gpurun_out/run24.log
cat gpurun_out/run24.log
