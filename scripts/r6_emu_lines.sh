# round 6: bench.py's own N > 1 lines with the ranks emulated on device 0 (what the driver's `--gpus N` would print, plus "multi_gpu_model": {"predicted": true, ...})
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for n in 2 8; do
  d=$(python -c "print(','.join(['0']*$n))")
  BFC_BENCH_DEVICES=$d timeout 900 python bench.py --gpus $n --steps 3 --warmup 1 > gpurun_out/r6_bench_gpus${n}_emulated.json 2> gpurun_out/r6_bench_gpus${n}_emulated.log; echo "--gpus $n rc=$?"
  python - gpurun_out/r6_bench_gpus${n}_emulated.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(d['n_gpus'], d['value'], d['ms_per_step'], d['scaling'], 'verified', d.get('verified')); print(json.dumps(d.get('multi_gpu_model'))[:900])
PY
done
