# round 6: what the default GPU suite no longer runs, on the final build: the variants behind BFC_TEST_MORE=1, two other seed bases of every fuzz family, other damaged files through the drop-in binaries
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
{
echo "== BFC_TEST_MORE=1: the variants taken out of the default run"
BFC_TEST_MORE=1 timeout 900 python -m pytest tests/test_gpu_group.py tests/test_gpu_parity.py -q -m gpu -x -k "sizes_stay_on_the_device or push_kernel or level1_slab_overflow or largest_filter_b37" 2>&1 | tail -2
echo "== build"; python -c "from bfc_amd import _lib; print(_lib.build_id())"
} > gpurun_out/r6_more.txt 2>&1
bash scripts/more_fuzz.sh 2 3 >> gpurun_out/r6_more.txt 2>&1
MORE_DROPIN=1 bash scripts/more_fuzz.sh 2 >> gpurun_out/r6_more.txt 2>&1
grep -v "^\.\.\." gpurun_out/r6_more.txt | tail -20
