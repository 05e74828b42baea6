# round 6: GPU evidence runs.  PARTS selects among: multi (first thing on any lease: bench.py --gpus 2 if the box has two devices), tests (whole GPU
# suite with durations + smoke), quick (fuzz + parity + group), sel (pytest -k "$SEL"), bench (driver-style line), ab (c3/c4e bench under ENV_A / ENV_B),
# emu (ranks emulated on one device: the work-inflation table), k55 (the published command's geometry), prof / prof2 (traces + PMC)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export TMPDIR=/tmp
PARTS=${PARTS:-"quick bench"}
FILT="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl"
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
summ() { python - "$1" <<'PY'
import json, sys
try: d = json.load(open(sys.argv[1]))
except Exception as e: print("no JSON:", e); sys.exit()
if d.get("error"): print("ERROR", d["error"]); sys.exit()
print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], 'verified', d.get('verified'), d['build_id'], 'frac', d['roofline']['frac'], d['roofline']['whole_job_frac'], 'n_gpus', d['n_gpus'], 'lib batches', d['config']['library_batches_per_step'])
s = d.get('secondary', {})
for k in s: print(k, s[k].get('value'), s[k].get('ms_per_step', s[k].get('gpu_ms_per_pass')), 'verified', s[k].get('verified'), 'frac', (s[k].get('roofline') or {}).get('frac'), s[k].get('stage_ms_per_step'), s[k].get('error'))
print('cpu', (d.get('cpu_baseline') or {}).get('value'), 'e2e', (d.get('e2e') or {}).get('mkmers_per_s'), (d.get('e2e') or {}).get('all_runs_s'), 'pcie', (d.get('pcie') or {}).get('mkmers_per_s'))
for k in ('e2e_gz', 'e2e_pipe'):
    if d.get(k): print(k, d[k].get('all_runs_s'), d[k].get('error'))
if d.get('verified') is False: print(d.get('verification'))
for k in s:
    if s[k].get('verified') is False: print(k, s[k].get('verification'))
PY
}
# a lease with two devices is the first one in six rounds -- measure the exchange before anything else
NDEV=$(python -c "from bfc_amd import _lib; print(int(_lib.load().bfcg_device_count()))" 2>/dev/null || echo 0)
echo "HIP devices on this box: $NDEV"
if [ "$NDEV" -ge 2 ]; then
  for n in 2 4 8; do
    if [ "$NDEV" -ge $n ]; then
      timeout 1200 python bench.py --gpus $n --steps 5 --warmup 2 > gpurun_out/r6_bench_gpus$n.json 2> gpurun_out/r6_bench_gpus$n.log; echo "--gpus $n rc=$?"; summ gpurun_out/r6_bench_gpus$n.json
    fi
  done
fi
if has quick; then
  timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_group.py -q -m gpu -x 2>&1 | grep -v "$FILT" > gpurun_out/r6_quick.log; tail -6 gpurun_out/r6_quick.log
fi
if has sel; then
  timeout ${SEL_TIMEOUT:-1500} python -m pytest ${SEL_FILES:-tests} -q -m gpu -x -k "$SEL" --durations=15 2>&1 | grep -v "$FILT" > gpurun_out/r6_sel.log; tail -${SEL_TAIL:-30} gpurun_out/r6_sel.log
fi
if has tests; then
  timeout 2400 python -m pytest tests -q -m gpu -x --durations=60 2>&1 | grep -v "$FILT" > gpurun_out/r6_gpu_tests.log; tail -75 gpurun_out/r6_gpu_tests.log
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
fi
if has bench; then
  timeout 2400 python bench.py --steps ${STEPS:-10} --warmup 3 ${BENCH_ARGS:-} > gpurun_out/r6_bench.json 2> gpurun_out/r6_bench.log; echo bench rc=$?
  summ gpurun_out/r6_bench.json
fi
if has ab; then
  for v in A B ${AB_MORE:-}; do
    eval "envs=\$ENV_$v"; eval "xargs_=\$ARGS_$v"
    env ${envs:-X=1} timeout 900 python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-boundary ${AB_ARGS:---no-secondary} $xargs_ > gpurun_out/r6_ab_$v.json 2> gpurun_out/r6_ab_$v.log; echo "ab $v ($envs $xargs_) rc=$?"
    summ gpurun_out/r6_ab_$v.json
  done
fi
if has e2e; then  # the boundary under the parser's phase clocks (BFC_INGEST_TIMING): the whole c3 file through bfc-dropin, from the file and from a pipe
  python - <<PY
import sys, time; sys.path.insert(0,'.')
from bfc_amd import gen
t=time.time(); rs = gen.ReadSet(seed=3, G=248_000_000, cov=30)
rs.fastq_parallel('/dev/shm/c3e.fq', 0, min(${READS:-49600000}, rs.n_reads), threads=32); print('written in %.1f s' % (time.time()-t))
PY
  export BFC_GPU_TIMING=1
  for t in ${E2E_TS:-64 64 32}; do echo "== file, -t$t"; ( time BFC_INGEST_TIMING=1 oracle/_ref/bfc-dropin -E -s 250m -k 33 -t$t /dev/shm/c3e.fq ) 2>&1 | grep -E "^real|T::|Real time" | tail -60; sleep 2; done > gpurun_out/r6_e2e.txt 2>&1
  for t in 64 64; do echo "== pipe, -t$t"; ( time sh -c "cat /dev/shm/c3e.fq | oracle/_ref/bfc-dropin -E -s 250m -k 33 -t$t -" ) 2>&1 | grep -E "^real|T::bfc|Real time" | tail -12; sleep 2; done >> gpurun_out/r6_e2e.txt 2>&1
  echo "== pipe, old serial path (BFC_INGEST_NO_PIPE=1), first 8 M reads" >> gpurun_out/r6_e2e.txt
  ( time sh -c "head -c 2500000000 /dev/shm/c3e.fq | BFC_INGEST_NO_PIPE=1 oracle/_ref/bfc-dropin -E -s 250m -k 33 -t64 -" ) 2>&1 | grep -E "^real|T::bfc_count\] waited|Real time" >> gpurun_out/r6_e2e.txt
  rm -f /dev/shm/c3e.fq; grep -E "==|Real time|^real|waited" gpurun_out/r6_e2e.txt; grep "T::fq" gpurun_out/r6_e2e.txt | head -12
fi
if has emu; then  # the predicted scaling table: ranks emulated on this one device
  timeout 2400 python scripts/mg_predict.py ${EMU_W:-c3 c4e} > gpurun_out/round6_mg_predicted.md 2> gpurun_out/r6_emu.log; echo "emu rc=$?"; cat gpurun_out/round6_mg_predicted.md; tail -3 gpurun_out/r6_emu.log
fi
if has prof; then  # traces + PMC passes of the headline workload on THIS build (bench.py's roofline.traffic reads profiles/round6_c3_pmc.json)
  PMC=2 STEPS=1 bash scripts/prof_round2.sh c3 > gpurun_out/prof_c3.out 2>&1; tail -3 gpurun_out/prof_c3.out | cut -c1-200
  ROUND=6 python tools/make_round_md.py gpurun_out/prof_c3 c3 > gpurun_out/round6_c3.md; cp profiles/round6_c3_pmc.json gpurun_out/ 2>/dev/null; head -20 gpurun_out/round6_c3.md | cut -c1-220
fi
if has prof2; then  # ... and of the secondaries the line carries
  for w in ${PROF2_W:-c4e c2}; do
    PMC=1 STEPS=${PROF2_STEPS:-2} BENCH_ARGS="--workload $w" bash scripts/prof_round2.sh $w > gpurun_out/prof_$w.out 2>&1; tail -2 gpurun_out/prof_$w.out | cut -c1-200
    ROUND=6 python tools/make_round_md.py gpurun_out/prof_$w $w > gpurun_out/round6_$w.md; cp profiles/round6_${w}_pmc.json gpurun_out/ 2>/dev/null
  done
fi
if has proftrim; then  # config c5's query pass as bench.py runs it (count pass of c5e, then the trim leg), under the kernel trace
  export TMPDIR=/tmp; mkdir -p gpurun_out/prof_c5e_trim
  timeout -k 5 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c5e_trim/t -o p -- python bench.py --workload c5e --steps 1 --warmup 0 --no-cpu-baseline --no-boundary --no-secondary > gpurun_out/prof_c5e_trim/t.log 2>&1; echo "proftrim rc=$?"
  python tools/rocpd_stats.py gpurun_out/prof_c5e_trim/t/p_results.db | cut -c1-170 | head -14 | tee gpurun_out/round6_c5e_trim_kernels.txt
  grep '"metric"' gpurun_out/prof_c5e_trim/t.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d.get('trim')))" | tee -a gpurun_out/round6_c5e_trim_kernels.txt | cut -c1-600
fi
if has d2h; then  # the table's way back with more copy threads (BFC_GPU_D2H_THREADS; default 8): the whole c3 file through bfc-dropin
  python - <<PY
import sys; sys.path.insert(0,'.')
from bfc_amd import gen
rs = gen.ReadSet(seed=3, G=248_000_000, cov=30)
rs.fastq_parallel('/dev/shm/c3e.fq', 0, rs.n_reads, threads=32)
PY
  export BFC_GPU_TIMING=1
  for t in 8 16 24 8 16 24; do echo "== BFC_GPU_D2H_THREADS=$t"; BFC_GPU_D2H_THREADS=$t oracle/_ref/bfc-dropin -E -s 250m -k 33 -t64 /dev/shm/c3e.fq 2>&1 | grep -E "Real time|export_table|left " | cut -c1-200; sleep 2; done > gpurun_out/r6_d2h.txt 2>&1
  rm -f /dev/shm/c3e.fq; cat gpurun_out/r6_d2h.txt
fi
if has morefuzz; then  # other draws of every fuzz family on this build (the suite itself runs fewer per family since round 6)
  bash scripts/more_fuzz.sh ${FUZZ_BASES:-1} > gpurun_out/r6_more_fuzz.txt 2>&1; cat gpurun_out/r6_more_fuzz.txt | tail -6
fi
