cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/exp9
O=gpurun_out/exp9
timeout 300 build/commit_probe 18 11 1700 4 > $O/commit_probe_c3.txt 2>&1; tail -11 $O/commit_probe_c3.txt | cut -c1-220
timeout 300 build/commit_probe 18 11 1700 1 > $O/commit_probe_c3_1p.txt 2>&1; tail -5 $O/commit_probe_c3_1p.txt | cut -c1-220
