cd $GRAFT_REPO_ROOT
timeout 600 python scripts/pcie_rate.py 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3 | tee gpurun_out/round2_pcie_rate.txt
