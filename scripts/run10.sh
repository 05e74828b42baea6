cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
{
echo "== single"; python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-1200
echo "== torchrun world=1 forced dist"; BFC_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -4 | cut -c1-1500
} > gpurun_out/run10.log 2>&1
cat gpurun_out/run10.log
