cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/run22.log
python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['roofline']['frac'], d['config']['slow_buckets'], d['config']['tab_cshift'])
" >> gpurun_out/run22.log 2>&1
timeout 900 python scripts/c3_run.py --b 35 --batch-reads 1572864,2097152 >> gpurun_out/run22.log 2>&1
cat gpurun_out/run22.log
