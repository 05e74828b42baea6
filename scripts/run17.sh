cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
{
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for cov in 25 100; do
echo "== single cov $cov"; python bench.py --steps 2 --warmup 1 --no-cpu-baseline --cov $cov 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['n_seen'], d['config']['n_distinct'], d['config']['batches_per_step'], d['config']['slow_buckets'], d['config']['stage_ms_per_step'], d['roofline']['frac'])"
echo "== dist cov $cov"; BFC_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --cov $cov 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['n_seen'], d['config']['n_distinct'], d['config']['batches_per_step'], d['config']['slow_buckets'], d['config']['stage_ms_per_step'])"
done
} > gpurun_out/run17.log 2>&1; cat gpurun_out/run17.log
