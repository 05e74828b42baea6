cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/run12.log 2>&1; tail -12 gpurun_out/run12.log | cut -c1-600
