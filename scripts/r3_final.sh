# round 3 evidence: c4 / c5 at full size, rocprofv3 traces + PMC passes of c3 and c2 (summaries -> gpurun_out/, copied to profiles/ by hand)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout 900 python scripts/c4_run.py --batch-reads 16777216 > gpurun_out/r3_c4_16m.log 2>&1; tail -4 gpurun_out/r3_c4_16m.log | cut -c1-400
timeout 900 python scripts/c4_run.py --batch-reads 8388608 --filter-mode 1 --k 51 --trim 1 > gpurun_out/r3_c5_8m.log 2>&1; tail -4 gpurun_out/r3_c5_8m.log | cut -c1-400
PMC=2 STEPS=1 bash scripts/prof_round2.sh c3 > gpurun_out/prof_c3.out 2>&1; tail -3 gpurun_out/prof_c3.out | cut -c1-200
PMC=2 STEPS=3 BENCH_ARGS="--workload c2" bash scripts/prof_round2.sh c2 > gpurun_out/prof_c2.out 2>&1; tail -3 gpurun_out/prof_c2.out | cut -c1-200
ROUND=3 python tools/make_round_md.py gpurun_out/prof_c3 c3 > gpurun_out/round3_c3.md; cp profiles/round3_c3_pmc.json gpurun_out/
ROUND=3 python tools/make_round_md.py gpurun_out/prof_c2 c2 > gpurun_out/round3_c2.md; cp profiles/round3_c2_pmc.json gpurun_out/
