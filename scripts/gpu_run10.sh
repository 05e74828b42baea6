cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -x -k "not c3_full and not c4_param and not c5_param" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-verify"
show() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], d['value'], d['ms_per_step'], d['config']['batch_reads'], d['config']['library_batches_per_step'], d['config']['slow_buckets'], d['config']['stage_ms_per_step'])
if 'secondary' in d: print(d['secondary']['c2']['value'], d['secondary']['c2']['stage_ms_per_step'], d['secondary']['c2'].get('verified'))
PY
}
$B > gpurun_out/x_a.json 2>/dev/null; show gpurun_out/x_a.json
BFCG_REC_DROP=0 $B --no-secondary > gpurun_out/x_b.json 2>/dev/null; show gpurun_out/x_b.json
BFCG_SYNC_BATCHES=1 $B --no-secondary > gpurun_out/x_c.json 2>/dev/null; show gpurun_out/x_c.json
