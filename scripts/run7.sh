cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
for cfg in "BFCG_ABLATE=64" "BFCG_ABLATE=64 BFCG_R=8 BFCG_LDS=53000 BFCG_AG=256 BFCG_BT=512"; do
  for br in 524288 786432; do
    echo "== $cfg $br"
    env $cfg python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch-reads $br 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['config']['phase_cycles'], d['config']['slow_buckets'])"
  done
done
} > gpurun_out/run7.log 2>&1
cat gpurun_out/run7.log
