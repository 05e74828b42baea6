cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/exp7
export TMPDIR=/tmp
O=gpurun_out/exp7
cat /sys/kernel/mm/transparent_hugepage/enabled
bash scripts/e2e_c3.sh > $O/e2e_c3.txt 2>&1; grep -E "==|Real time|^real|device buffers|clean-up|waited" $O/e2e_c3.txt | cut -c1-260
