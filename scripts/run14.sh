cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
{ python scripts/trim_rate.py 51 33; python scripts/trim_rate.py 51 37; } > gpurun_out/run14.log 2>&1; cat gpurun_out/run14.log
