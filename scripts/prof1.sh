cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof1 -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof1/bench.log 2>&1
ls -R gpurun_out/prof1 | head -30
f=$(find gpurun_out/prof1 -name "*kernel_stats.csv" | head -1); echo $f; cat $f | head -30
