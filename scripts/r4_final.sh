# round 4 evidence run of a build: GPU suite, smoke, the driver-style bench line, the group path at world size 1, c4 / c5 at full size,
# rocprofv3 traces + PMC passes of c3 and c2 (summaries -> gpurun_out/, copied to profiles/ by hand).  PARTS="tests bench dist c4 c5 prof" selects.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export TMPDIR=/tmp
PARTS=${PARTS:-"tests bench dist c4 c5 prof e2e"}
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
if has tests; then
  timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > gpurun_out/r4_gpu_tests.log; tail -8 gpurun_out/r4_gpu_tests.log
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
fi
if has bench; then
  timeout 1500 python bench.py --steps ${STEPS:-10} --warmup 3 > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.log; echo bench rc=$?
  python - <<'PY'
import json
d=json.load(open('gpurun_out/r4_bench.json'))
print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], 'verified', d.get('verified'), d['build_id'], 'frac', d['roofline']['frac'], d['roofline']['whole_job_frac'])
s=d.get('secondary',{})
for k in s: print(k, s[k].get('value'), s[k].get('ms_per_step'), s[k].get('verified'), s[k].get('stage_ms_per_step'), s[k].get('error'))
print('cpu', (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('cpu_model'), 'e2e', (d.get('e2e') or {}).get('mkmers_per_s'), 'pcie', (d.get('pcie') or {}).get('mkmers_per_s'))
PY
fi
if has dist; then
  BFC_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-boundary > gpurun_out/r4_bench_dist1.json 2> gpurun_out/r4_bench_dist1.log
  python -c "
import json; d=json.load(open('gpurun_out/r4_bench_dist1.json')); print('dist1', d['value'], d['ms_per_step'], d.get('verified'), d['config']['parallelism'][:80])"
fi
if has c4; then timeout 900 python scripts/c4_run.py --batch-reads 16777216 > gpurun_out/r4_c4_16m.log 2>&1; tail -2 gpurun_out/r4_c4_16m.log | cut -c1-700; fi
if has c5; then timeout 900 python scripts/c4_run.py --batch-reads 8388608 --filter-mode 1 --k 51 --trim 1 > gpurun_out/r4_c5_8m.log 2>&1; tail -3 gpurun_out/r4_c5_8m.log | cut -c1-700; fi
if has prof; then
  PMC=2 STEPS=1 bash scripts/prof_round2.sh c3 > gpurun_out/prof_c3.out 2>&1; tail -3 gpurun_out/prof_c3.out | cut -c1-200
  PMC=2 STEPS=3 BENCH_ARGS="--workload c2" bash scripts/prof_round2.sh c2 > gpurun_out/prof_c2.out 2>&1; tail -3 gpurun_out/prof_c2.out | cut -c1-200
  ROUND=4 python tools/make_round_md.py gpurun_out/prof_c3 c3 > gpurun_out/round4_c3.md; cp profiles/round4_c3_pmc.json gpurun_out/
  ROUND=4 python tools/make_round_md.py gpurun_out/prof_c2 c2 > gpurun_out/round4_c2.md; cp profiles/round4_c2_pmc.json gpurun_out/
  PROF_TIMEOUT=400 bash scripts/prof_sq.sh r4final > gpurun_out/r4_prof_sq.log 2>&1
fi
if has e2e; then READS=49600000 PLANES="1 0 1" bash scripts/e2e_c3.sh > gpurun_out/r4_e2e_c3_full.txt 2>&1; grep -E "==|Real time|^real|written" gpurun_out/r4_e2e_c3_full.txt | cut -c1-200; fi
