cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -x -k "not c3_full and not c4_param and not c5_param" 2>&1 | tail -6
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-verify"
show() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], d['value'], d['ms_per_step'], d['config']['batch_reads'], d['config']['library_batches_per_step'], d['config']['slow_buckets'], d['config']['stage_ms_per_step'])
if 'secondary' in d: print(d['secondary']['c2']['value'], d['secondary']['c2']['stage_ms_per_step'], d['secondary']['c2'].get('verified'))
PY
}
$B > gpurun_out/x_a.json 2>/dev/null; show gpurun_out/x_a.json
for br in 1048576 2097152 4194304 8388608; do $B --no-secondary --batch-reads $br > gpurun_out/x_br$br.json 2>/dev/null; show gpurun_out/x_br$br.json; done
gen() { python - <<'PY'
import sys; sys.path.insert(0,'.')
from bfc_amd import gen
gen.fixture('c2').fastq('/dev/shm/c2.fq')
PY
}
gen; ls -la /dev/shm/c2.fq
for i in 1 2; do /usr/bin/time -f "dropin c2 wall %e s" env BFC_GPU_TIMING=1 oracle/_ref/bfc-dropin -E -k31 -t32 /dev/shm/c2.fq 2>&1 | grep -v "^\[M::bfc_count_cb\] read\|processed" | tail -6; done
rm -f /dev/shm/c2.fq
