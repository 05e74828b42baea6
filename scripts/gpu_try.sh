cd $GRAFT_REPO_ROOT
for v in "X=1" "BFCG_ONEPASS=0" "BFCG_NO_WARM_BATCHES=1" "BFCG_PIPELINE=0" "BFCG_SEG=0"; do
echo "== $v"; env $v BFC_FUZZ_SEED_BASE=21 timeout 300 python -m pytest "tests/test_gpu_fuzz.py::test_random_configuration_on_emulated_ranks[0]" -q -m gpu -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -1
done
