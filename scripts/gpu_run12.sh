cd $GRAFT_REPO_ROOT
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-secondary"
show() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], d['value'], d['ms_per_step'], d['config']['batch_reads'], d['config']['library_batches_per_step'], d['config']['slow_buckets'], d['config']['stage_ms_per_step'])
PY
}
$B > gpurun_out/x_a.json 2>/dev/null; show gpurun_out/x_a.json
BFCG_COLD_FRAC=0.5 $B > gpurun_out/x_b.json 2>/dev/null; show gpurun_out/x_b.json
BFCG_COLD_FRAC=0.34 $B > gpurun_out/x_c.json 2>/dev/null; show gpurun_out/x_c.json
BFCG_COLD_FRAC=0.25 $B > gpurun_out/x_d.json 2>/dev/null; show gpurun_out/x_d.json
