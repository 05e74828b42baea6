cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
{
python - <<'PY'
import sys; sys.path.insert(0,'.')
from bfc_amd import gen
rs = gen.ReadSet(seed=2, G=4_600_000, cov=100)
rs.fastq('/dev/shm/c2.fq'); print('reads', rs.n_reads)
PY
ls -la /dev/shm/c2.fq
echo "== dropin -E (GPU count, file ingest)"; ( time oracle/_ref/bfc-dropin -E -k31 /dev/shm/c2.fq ) 2>&1 | grep -E "Real time|real|distinct" | tail -3
echo "== dropin -E again"; ( time oracle/_ref/bfc-dropin -E -k31 /dev/shm/c2.fq ) 2>&1 | grep -E "Real time|real" | tail -2
echo "== ref -E -t256"; ( time oracle/_ref/bfc-ref -E -k31 -t256 /dev/shm/c2.fq ) 2>&1 | grep -E "Real time|real" | tail -2
echo "== ref -E -t64"; ( time oracle/_ref/bfc-ref -E -k31 -t64 /dev/shm/c2.fq ) 2>&1 | grep -E "Real time|real" | tail -2
rm -f /dev/shm/c2.fq
} > gpurun_out/e2e.log 2>&1; cat gpurun_out/e2e.log
