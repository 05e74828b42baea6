cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -x -k "not c4_param and not c5_param" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4
python scripts/bloom_phases.py 2>&1 | grep batch | head -4
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-verify"
show() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], d['value'], d['ms_per_step'], d['config']['batch_reads'], d['config']['library_batches_per_step'], d['config']['slow_buckets'], d['config']['stage_ms_per_step'])
if 'secondary' in d: print(d['secondary']['c2']['value'], d['secondary']['c2']['stage_ms_per_step'], d['secondary']['c2'].get('verified'))
PY
}
$B > gpurun_out/x_a.json 2>/dev/null; show gpurun_out/x_a.json
