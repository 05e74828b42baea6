cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_baseline_shapes.py -q -m gpu -x -k "gz or skewed" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4
bash scripts/more_fuzz.sh 11 12 13 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -9
MORE_DROPIN=1 bash scripts/more_fuzz.sh 5 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
