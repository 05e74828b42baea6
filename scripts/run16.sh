cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
python scripts/mgcheck.py > gpurun_out/run16.log 2>&1; cat gpurun_out/run16.log
