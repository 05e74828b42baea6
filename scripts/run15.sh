cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
{ python scripts/k33_rate.py 31 33; python scripts/k33_rate.py 33 33; python scripts/k33_rate.py 33 35; python scripts/k33_rate.py 33 37; python scripts/k33_rate.py 51 33; python scripts/k33_rate.py 63 35; } > gpurun_out/run15.log 2>&1; cat gpurun_out/run15.log
