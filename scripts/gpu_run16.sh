cd $GRAFT_REPO_ROOT
timeout 900 python scripts/c4_run.py --batch-reads 8388608 > gpurun_out/r2_c4_8m.log 2>&1; tail -2 gpurun_out/r2_c4_8m.log | cut -c1-1500
timeout 900 python scripts/c4_run.py --batch-reads 16777216 > gpurun_out/r2_c4_16m.log 2>&1; tail -2 gpurun_out/r2_c4_16m.log | cut -c1-1500
