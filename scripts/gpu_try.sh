cd $GRAFT_REPO_ROOT
timeout 230 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_parity.py tests/test_kcov.py -q -m gpu -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -2
