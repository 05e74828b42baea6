cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
{
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for cfg in "BFCG_ABLATE=0" "BFCG_LDS=39000" "BFCG_R=7 BFCG_LDS=39000 BFCG_AG=128" "BFCG_R=7 BFCG_LDS=26000 BFCG_AG=128 BFCG_BT=256" "BFCG_BT=256 BFCG_LDS=39000"; do
  for br in 786432 1048576 1572864; do
    echo "== $cfg $br"
    env $cfg python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch-reads $br 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['config']['slow_buckets'])"
  done
done
} > gpurun_out/run11.log 2>&1
cat gpurun_out/run11.log
