# end to end from a gzip file: the reference's main() on this library (bfc-dropin) vs the reference binary, both with -t32, c2-sized input (25x here: gzip -6 of 100x takes minutes)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
{
python - <<'PY'
import sys; sys.path.insert(0,'.')
from bfc_amd import gen
rs = gen.ReadSet(seed=2, G=4_600_000, cov=50)
rs.fastq('/dev/shm/c2h.fq'); print('reads', rs.n_reads)
PY
( time gzip -6 -k -f /dev/shm/c2h.fq ) 2>&1 | grep real
ls -l /dev/shm/c2h.fq /dev/shm/c2h.fq.gz | awk '{print $5, $9}'
export BFC_GPU_TIMING=1
for f in /dev/shm/c2h.fq /dev/shm/c2h.fq.gz; do
  echo "== bfc-dropin -E -k31 -t32 $f"; for i in 1 2; do ( time oracle/_ref/bfc-dropin -E -k31 -t32 $f ) 2>&1 | grep -E "^real|T::|Real time" ; done
done
unset BFC_GPU_TIMING
echo "== reference bfc -E -k31 -t32 on the .gz"; ( time oracle/_ref/bfc-ref -E -k31 -t32 /dev/shm/c2h.fq.gz ) 2>&1 | grep -E "^real|Real time"
oracle/_ref/bfc-dropin -E -k31 -t32 -d /dev/shm/a.hash /dev/shm/c2h.fq.gz 2>/dev/null; oracle/_ref/bfc-dropin -E -k31 -t1 -d /dev/shm/b.hash /dev/shm/c2h.fq 2>/dev/null
python - <<'PY'
import sys; sys.path.insert(0,'.')
import oracle
a = oracle.parse_dump('/dev/shm/a.hash'); b = oracle.parse_dump('/dev/shm/b.hash')
print('L1 digest of the table from the .gz (32 inflating threads):', oracle.l1_digest(a[2], a[3]), ' from the plain file (serial parser):', oracle.l1_digest(b[2], b[3]))
PY
rm -f /dev/shm/c2h.fq /dev/shm/c2h.fq.gz /dev/shm/a.hash /dev/shm/b.hash
} > gpurun_out/round2_e2e_gz.txt 2>&1; cat gpurun_out/round2_e2e_gz.txt
