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
