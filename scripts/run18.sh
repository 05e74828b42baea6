cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout 300 python scripts/a2a_probe.py > gpurun_out/run18.log 2>&1; tail -8 gpurun_out/run18.log
