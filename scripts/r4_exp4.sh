cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/exp4
export TMPDIR=/tmp
O=gpurun_out/exp4
bash scripts/e2e_c3.sh > $O/e2e_c3.txt 2>&1; tail -30 $O/e2e_c3.txt | cut -c1-300
timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > $O/gpu_tests.log; tail -5 $O/gpu_tests.log
