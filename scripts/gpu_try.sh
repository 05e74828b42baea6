cd $GRAFT_REPO_ROOT
python scripts/op_debug.py 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl"
timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8
