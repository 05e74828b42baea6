cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
{
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for cfg in "BFCG_SYNC_BATCHES=1" "BFCG_ABLATE=0"; do
    echo "== $cfg"
    env $cfg python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['config']['slow_buckets'], d['roofline'])"
done
} > gpurun_out/run11.log 2>&1
cat gpurun_out/run11.log
