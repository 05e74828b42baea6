# round 6: configs c4 and c5 (count + trim) at FULL size on one MI355X, final build
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout 1200 python scripts/c4_run.py --batch-reads 16777216 > gpurun_out/r6_c4_16m.log 2>&1; echo "c4 rc=$?"; tail -1 gpurun_out/r6_c4_16m.log | cut -c1-900
timeout 1200 python scripts/c4_run.py --batch-reads 16777216 --filter-mode 1 --k 51 --trim 1 > gpurun_out/r6_c5_16m.log 2>&1; echo "c5 rc=$?"; tail -1 gpurun_out/r6_c5_16m.log | cut -c1-900
