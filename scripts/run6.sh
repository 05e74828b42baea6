cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8
for cfg in "BFCG_ABLATE=0" "BFCG_INLINE_COMMIT=1" "BFCG_BT=256" ; do
  for br in 524288 ; do
    echo "== $cfg batch_reads=$br"
    env $cfg python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch-reads $br 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['config']['slow_buckets'], d['roofline']['frac'])"
  done
done
} > gpurun_out/run6.log 2>&1
cat gpurun_out/run6.log
