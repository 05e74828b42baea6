cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout 900 python scripts/c3_run.py --b 35 --batch-reads 1048576,1572864,2097152 > gpurun_out/run21.log 2>&1
cat gpurun_out/run21.log
