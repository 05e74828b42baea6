cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/exp3
export TMPDIR=/tmp
O=gpurun_out/exp3
timeout 300 build/commit_probe 17 13 1335 2 > $O/commit_probe_2p.txt 2>&1; tail -11 $O/commit_probe_2p.txt | cut -c1-200
bash scripts/e2e_c3.sh > $O/e2e_c3.txt 2>&1; tail -24 $O/e2e_c3.txt | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_dropin.py -q -x -m gpu 2>&1 | tail -3
