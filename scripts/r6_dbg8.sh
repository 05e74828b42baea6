cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in "X=1"; do
  env $v BFC_BENCH_DEVICES=0,0,0,0,0,0,0,0 timeout 900 python bench.py --gpus 8 --workload c4e --batch-reads 8388608 --steps 1 --warmup 0 --no-cpu-baseline --no-boundary --no-secondary > gpurun_out/dbg8_$v.json 2> gpurun_out/dbg8_$v.log
  echo "== $v rc=$?"; python - "gpurun_out/dbg8_$v.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(d.get('verified'), d.get('ms_per_step'), d['config']['partition'], d['config']['library_batches_per_step']); print({k:v for k,v in (d.get('verification') or {}).items() if isinstance(v,dict) or k in ('error',)})
PY
  grep -i "D::group\|overflow\|two-pass\|W::" gpurun_out/dbg8_$v.log | head -5
done
