cd $GRAFT_REPO_ROOT
show() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], d['value'], d['ms_per_step'], d['config']['batch_reads'], d['config']['library_batches_per_step'], d['config']['slow_buckets'], d['config']['stage_ms_per_step'])
PY
}
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-secondary"
for br in 4194304 3670016; do $B --batch-reads $br > gpurun_out/x_br$br.json 2>/dev/null; show gpurun_out/x_br$br.json; done
timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_group.py -q -m gpu -x 2>&1 | tail -3
python - <<'PY'
import sys; sys.path.insert(0,'.')
from bfc_amd import gen
gen.fixture('c2').fastq('/dev/shm/c2.fq')
PY
for i in 1 2 3; do ( time env BFC_GPU_TIMING=1 oracle/_ref/bfc-dropin -E -k31 -t32 /dev/shm/c2.fq ) 2>&1 | grep "T::\|Real time\|real" | tail -5; done
( time oracle/_ref/bfc-ref -E -k31 -t256 /dev/shm/c2.fq ) 2>&1 | grep "Real time\|real"
rm -f /dev/shm/c2.fq
