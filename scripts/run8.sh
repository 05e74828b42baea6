cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
{
BFC_BENCH_VERBOSE=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "batch|value" | cut -c1-300
} > gpurun_out/run8.log 2>&1
cat gpurun_out/run8.log
