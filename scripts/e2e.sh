cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
{
python - <<'PY'
import sys; sys.path.insert(0,'.')
from bfc_amd import gen
rs = gen.ReadSet(seed=2, G=4_600_000, cov=100)
rs.fastq('/dev/shm/c2.fq'); print('reads', rs.n_reads)
PY
export BFC_GPU_TIMING=1
for t in 1 32; do
  echo "== dropin -E -t$t (GPU count, file ingest)"; for i in 1 2; do ( time oracle/_ref/bfc-dropin -E -k31 -t$t /dev/shm/c2.fq ) 2>&1 | grep -E "^real|T::|Real time" ; done
done
unset BFC_GPU_TIMING
BFC_GPU_EXACT_DUMP=1 oracle/_ref/bfc-dropin -E -k31 -t32 -d /dev/shm/a.hash /dev/shm/c2.fq 2>/dev/null; BFC_GPU_EXACT_DUMP=1 oracle/_ref/bfc-dropin -E -k31 -t1 -d /dev/shm/b.hash /dev/shm/c2.fq 2>/dev/null; cmp /dev/shm/a.hash /dev/shm/b.hash && echo "dump identical for -t32 (fast ingest) and -t1 (serial ingest)"; md5sum /dev/shm/a.hash
rm -f /dev/shm/c2.fq /dev/shm/a.hash /dev/shm/b.hash
} > gpurun_out/e2e.log 2>&1; cat gpurun_out/e2e.log
