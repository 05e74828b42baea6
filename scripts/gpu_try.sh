cd $GRAFT_REPO_ROOT
timeout 600 python scripts/cliff_debug.py 2097152 4194304 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -12
