cd $GRAFT_REPO_ROOT
timeout 900 python scripts/c4_run.py --batch-reads 16777216 > gpurun_out/r2_c4_16m.log 2>&1; tail -1 gpurun_out/r2_c4_16m.log | cut -c1-1500
