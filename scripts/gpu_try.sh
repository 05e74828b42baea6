cd $GRAFT_REPO_ROOT
BFCG_ABLATE=64 BR=8388608 NB=6 timeout 600 python scripts/bloom_phases.py 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -12
