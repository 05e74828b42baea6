cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
{
BFCG_V4=1 BFCG_BT=1024 BFCG_LDS=79000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
for cfg in "BFCG_ABLATE=0" "BFCG_V4=1 BFCG_BT=1024 BFCG_LDS=79000" "BFCG_V4=1 BFCG_BT=512 BFCG_LDS=54300" "BFCG_V4=1 BFCG_BT=1024 BFCG_LDS=79000 BFCG_SYNC_BATCHES=1 BFCG_ABLATE=64"; do
  for br in 786432 1048576; do
    echo "== $cfg $br"
    env $cfg python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch-reads $br 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['config']['slow_buckets'], d['config']['phase_cycles'])"
  done
done
} > gpurun_out/run11.log 2>&1
cat gpurun_out/run11.log
