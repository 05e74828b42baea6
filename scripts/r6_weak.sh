cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
MG_SCALING=weak timeout 2400 python scripts/mg_predict.py c3 > gpurun_out/round6_mg_predicted_weak.md 2> gpurun_out/r6_weak.log; echo "rc=$?"; cat gpurun_out/round6_mg_predicted_weak.md | cut -c1-400; tail -5 gpurun_out/r6_weak.log | cut -c1-300
