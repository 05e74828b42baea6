# round 5: GPU evidence runs.  PARTS selects: quick (fuzz + parity + group tests), tests (whole GPU suite + smoke), bench (driver-style line),
# ab (c3 bench with ENV_A / ENV_B settings), inproc (bench.py --gpus 2 with ranks emulated on device 0, c3), c4 / c5 (full size), prof (traces + PMC)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export TMPDIR=/tmp
PARTS=${PARTS:-"quick bench"}
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
summ() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
if d.get("error"): print("ERROR", d["error"]); sys.exit()
print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], 'verified', d.get('verified'), d['build_id'], 'frac', d['roofline']['frac'], d['roofline']['whole_job_frac'], 'n_gpus', d['n_gpus'], 'lib batches', d['config']['library_batches_per_step'])
s = d.get('secondary', {})
for k in s: print(k, s[k].get('value'), s[k].get('ms_per_step'), s[k].get('verified'), s[k].get('stage_ms_per_step'), s[k].get('error'))
print('cpu', (d.get('cpu_baseline') or {}).get('value'), 'e2e', (d.get('e2e') or {}).get('mkmers_per_s'), (d.get('e2e') or {}).get('wall_s'), 'pcie', (d.get('pcie') or {}).get('mkmers_per_s'))
if d.get('verified') is False: print(d.get('verification'))
PY
}
if has quick; then
  timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_group.py -q -m gpu -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > gpurun_out/r5_quick.log; tail -6 gpurun_out/r5_quick.log
fi
if has fuzzenv; then  # the fuzz + parity suites under an environment setting (FUZZ_ENV="BFCG_X=1 ...")
  env $FUZZ_ENV timeout 1200 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > gpurun_out/r5_fuzzenv.log; echo "fuzzenv ($FUZZ_ENV)"; tail -4 gpurun_out/r5_fuzzenv.log
fi
if has tests; then
  timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > gpurun_out/r5_gpu_tests.log; tail -8 gpurun_out/r5_gpu_tests.log
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
fi
if has bench; then
  timeout 1800 python bench.py --steps ${STEPS:-10} --warmup 3 ${BENCH_ARGS:-} > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.log; echo bench rc=$?
  summ gpurun_out/r5_bench.json
fi
if has ab; then
  for v in A B; do
    eval "envs=\$ENV_$v"
    env $envs timeout 900 python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-boundary ${AB_ARGS:---no-secondary} > gpurun_out/r5_ab_$v.json 2> gpurun_out/r5_ab_$v.log; echo "ab $v ($envs) rc=$?"
    summ gpurun_out/r5_ab_$v.json
  done
fi
if has inproc; then
  BFC_BENCH_DEVICES=0,0 timeout 1200 python bench.py --gpus 2 --steps ${STEPS:-3} --warmup 1 > gpurun_out/r5_bench_gpus2_emulated.json 2> gpurun_out/r5_bench_gpus2_emulated.log; echo inproc rc=$?
  summ gpurun_out/r5_bench_gpus2_emulated.json
  python bench.py --gpus 2 --steps 1 --warmup 0 > gpurun_out/r5_bench_gpus2_plain.json 2>/dev/null; echo "plain --gpus 2 rc=$?"; cat gpurun_out/r5_bench_gpus2_plain.json
fi
if has c4; then timeout 900 python scripts/c4_run.py --batch-reads 16777216 > gpurun_out/r5_c4_16m.log 2>&1; tail -2 gpurun_out/r5_c4_16m.log | cut -c1-700; fi
if has c5; then timeout 900 python scripts/c4_run.py --batch-reads ${C5_BATCH:-16777216} --filter-mode 1 --k 51 --trim 1 > gpurun_out/r5_c5_${C5_BATCH:-16777216}.log 2>&1; tail -3 gpurun_out/r5_c5_${C5_BATCH:-16777216}.log | cut -c1-700; fi
if has prof; then
  PMC=2 STEPS=1 bash scripts/prof_round2.sh c3 > gpurun_out/prof_c3.out 2>&1; tail -3 gpurun_out/prof_c3.out | cut -c1-200
  ROUND=5 python tools/make_round_md.py gpurun_out/prof_c3 c3 > gpurun_out/round5_c3.md; cp profiles/round5_c3_pmc.json gpurun_out/ 2>/dev/null
fi
if has fm; then  # the filter-mode kernel: its parity tests, the c5s shape, the FM fuzz draws
  timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py tests/test_gpu_fuzz.py tests/test_gpu_group.py -q -m gpu -x -k "filter_mode or c5 or fm or fuzz or group_host" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > gpurun_out/r5_fm.log; tail -5 gpurun_out/r5_fm.log
fi
if has c5ab; then
  BFCG_NO_B3=1 timeout 900 python scripts/c4_run.py --batch-reads 8388608 --filter-mode 1 --k 51 --cov ${C5_COV:-8} > gpurun_out/r5_c5_old.log 2>&1; tail -1 gpurun_out/r5_c5_old.log | cut -c1-600
  timeout 900 python scripts/c4_run.py --batch-reads 8388608 --filter-mode 1 --k 51 --cov ${C5_COV:-8} > gpurun_out/r5_c5_new.log 2>&1; tail -1 gpurun_out/r5_c5_new.log | cut -c1-600
fi
if has trim; then  # the query kernel: trim tests, then c5's trim pass at 8x with and without k_query4
  timeout 1200 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_dropin.py -q -m gpu -x -k "trim" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > gpurun_out/r5_trim.log; tail -4 gpurun_out/r5_trim.log
  for q in 0 1; do
    BFCG_QUERY4=$q timeout 900 python scripts/c4_run.py --batch-reads 16777216 --filter-mode 1 --k 51 --cov ${C5_COV:-8} --trim 1 > gpurun_out/r5_c5_trim_q$q.log 2>&1; echo "BFCG_QUERY4=$q"; tail -1 gpurun_out/r5_c5_trim_q$q.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if k.startswith('trim') or k in ('gpu_s','gpu_stage_ms')})"
  done
fi
if has s1abl; then  # k_scatter1 under the measurement switches (the -DBFCG_MEASURE library): what each part costs on the current kernel
  for a in 0 2048 1024 256 512 1280 3328; do BFCG_ABLATE=$a timeout 300 python scripts/s1_ablate.py 2>&1 | tail -1; done > gpurun_out/r5_s1_ablate.txt; cat gpurun_out/r5_s1_ablate.txt
fi
if has group; then  # the group tests, then c3 through an in-process group of one (lazy sizes on / off) against the plain path, one box
  timeout 1500 python -m pytest tests/test_gpu_group.py -q -m gpu -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > gpurun_out/r5_group.log; tail -6 gpurun_out/r5_group.log
  for v in plain lazy1 lazy0; do
    case $v in plain) envs="X=1";; lazy1) envs="BFC_BENCH_FORCE_GROUP=1";; lazy0) envs="BFC_BENCH_FORCE_GROUP=1 BFCG_MG_LAZY=0";; esac
    env $envs timeout 900 python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-boundary --no-secondary > gpurun_out/r5_group_$v.json 2> gpurun_out/r5_group_$v.log; echo "group $v ($envs) rc=$?"
    summ gpurun_out/r5_group_$v.json
  done
fi
if has e2eab; then  # the boundary: bfc-dropin on the c3 FASTQ (READS of it) in tmpfs, the mapping given back behind the parser (default) against one munmap at the end
  python - <<PY
import sys, time; sys.path.insert(0,'.')
from bfc_amd import gen
t=time.time(); rs = gen.ReadSet(seed=3, G=248_000_000, cov=30)
rs.fastq_parallel('/dev/shm/c3e.fq', 0, min(${READS:-49600000}, rs.n_reads), threads=32); print('reads', min(${READS:-49600000}, rs.n_reads), 'written in %.1f s' % (time.time()-t))
PY
  ls -l /dev/shm/c3e.fq
  export BFC_GPU_TIMING=1
  for mm in 0 1 0 1; do echo "== unmapping behind the parser: $mm"; ( time BFC_INGEST_UNMAP_MIN=$((mm ? 268435456 : 1099511627776)) oracle/_ref/bfc-dropin -E -s 250m -k 33 -t${E2E_T:-64} /dev/shm/c3e.fq ) 2>&1 | grep -E "^real|T::|Real time" | tail -16; echo; done > gpurun_out/r5_e2e_ab.txt 2>&1
  rm -f /dev/shm/c3e.fq; grep -E "==|Real time|^real|clean-up|waited" gpurun_out/r5_e2e_ab.txt
fi
if has e2et; then  # the boundary with more parser threads (E2E_TS="64 128 256"), the whole c3 file
  python - <<PY
import sys, time; sys.path.insert(0,'.')
from bfc_amd import gen
rs = gen.ReadSet(seed=3, G=248_000_000, cov=30)
rs.fastq_parallel('/dev/shm/c3e.fq', 0, min(${READS:-49600000}, rs.n_reads), threads=32)
PY
  export BFC_GPU_TIMING=1
  for t in ${E2E_TS:-64 128 256 64 128}; do echo "== -t$t"; ( time oracle/_ref/bfc-dropin -E -s 250m -k 33 -t$t /dev/shm/c3e.fq ) 2>&1 | grep -E "^real|T::|Real time" | tail -16; echo; done > gpurun_out/r5_e2e_t.txt 2>&1
  rm -f /dev/shm/c3e.fq; grep -E "==|Real time|^real|waited" gpurun_out/r5_e2e_t.txt
fi
if has wc; then  # k_scatter1_wc: its fuzz family, then c3 with the tile kernel (BFCG_S1_WC=0) against the write-combining one (default), one box
  timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x -k "write_combining" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > gpurun_out/r5_wc_fuzz.log; tail -15 gpurun_out/r5_wc_fuzz.log
  for v in ${WC_AB:-BFCG_S1_WC=0 BFCG_S1_WC_BT=512 BFCG_S1_WC_BT=1024 BFCG_S1_WC_BT=512}; do
    env $v timeout 600 python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-boundary ${AB_ARGS:---no-secondary} > gpurun_out/r5_wc_$v.json 2> gpurun_out/r5_wc_$v.log; echo "$v rc=$?"
    summ gpurun_out/r5_wc_$v.json; tail -3 gpurun_out/r5_wc_$v.log | cut -c1-300
  done
fi
if has wcph; then  # k_scatter1_wc's phase clocks (the -DBFCG_MEASURE library, built before the call): cycles per round, P1 / P2
  for e in ${WCPH_ENVS:-X=1}; do env $e TAG=$e timeout 300 python scripts/s1wc_phases.py 2>&1 | tail -2; done | tee gpurun_out/r5_wc_phases.txt
fi
if has runs; then  # N more driver-style runs of c3 alone (no secondaries, no boundary, no CPU leg): the spread across a box's minutes
  for i in $(seq 1 ${RUNS:-4}); do
    timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-boundary --no-secondary > gpurun_out/r5_run_$i.json 2> gpurun_out/r5_run_$i.log; echo "run $i rc=$?"; summ gpurun_out/r5_run_$i.json
  done
fi
if has wc16; then  # k_scatter1_wc on 16-byte records: its fuzz family + the c5 shapes of the baseline tests, then c5 at 8x coverage with the tile kernel and with it
  timeout 1200 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_baseline_shapes.py tests/test_gpu_parity.py -q -m gpu -x -k "write_combining or c5 or filter_mode" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > gpurun_out/r5_wc16.log; tail -6 gpurun_out/r5_wc16.log
  for v in 0 1; do
    BFCG_S1_WC=$v timeout 900 python scripts/c4_run.py --batch-reads 16777216 --filter-mode 1 --k 51 --cov ${C5_COV:-8} > gpurun_out/r5_c5_wc$v.log 2>&1; echo "BFCG_S1_WC=$v"; tail -1 gpurun_out/r5_c5_wc$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('gpu_s','gpu_stage_ms','n_kmers','n_seen','partition')})"
  done
fi
if has prof2; then  # traces + PMC passes of the secondary workloads the bench line carries (so that their `traffic` is of this build): c2 and c4e
  for w in ${PROF2_W:-c2 c4e}; do
    PMC=1 STEPS=${PROF2_STEPS:-2} BENCH_ARGS="--workload $w" bash scripts/prof_round2.sh $w > gpurun_out/prof_$w.out 2>&1; tail -2 gpurun_out/prof_$w.out | cut -c1-200
    ROUND=5 python tools/make_round_md.py gpurun_out/prof_$w $w > gpurun_out/round5_$w.md; cp profiles/round5_${w}_pmc.json gpurun_out/ 2>/dev/null
  done
fi
if has e2e512; then  # the boundary with the parser's AVX-512 pack (default) against the eight-positions-per-step one (BFC_INGEST_NO_AVX512=1): the whole c3 file through bfc-dropin, three runs each
  python - <<PY
import sys; sys.path.insert(0,'.')
from bfc_amd import gen
rs = gen.ReadSet(seed=3, G=248_000_000, cov=30)
rs.fastq_parallel('/dev/shm/c3e.fq', 0, rs.n_reads, threads=32)
PY
  export BFC_GPU_TIMING=1
  for v in 0 1 0 1 0 1; do echo "== BFC_INGEST_NO_AVX512=$v"; ( if [ $v = 1 ]; then export BFC_INGEST_NO_AVX512=1; fi; oracle/_ref/bfc-dropin -E -s 250m -k 33 -t64 /dev/shm/c3e.fq ) 2>&1 | grep -E "Real time|waited for the parser" | cut -c1-200; done > gpurun_out/r5_e2e_avx512.txt 2>&1
  rm -f /dev/shm/c3e.fq; cat gpurun_out/r5_e2e_avx512.txt
fi
