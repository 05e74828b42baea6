# the boundary on config c3's shape: the reference's unmodified main() on libbfc_gpu.so from a FASTQ file in tmpfs, with the library's phase times
# READS (default 16777216, as bench.py's e2e leg; 49600000 = the whole c3 read set, 15.6 GB of FASTQ)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
READS=${READS:-16777216}
python - <<PY
import sys, time; sys.path.insert(0,'.')
from bfc_amd import gen
t=time.time(); rs = gen.ReadSet(seed=3, G=248_000_000, cov=30)
rs.fastq('/dev/shm/c3e.fq', 0, min($READS, rs.n_reads)); print('reads', min($READS, rs.n_reads), 'written in %.1f s' % (time.time()-t))
PY
ls -l /dev/shm/c3e.fq
export BFC_GPU_TIMING=1
for pl in ${PLANES:-1 0 1 0}; do echo "== BFC_GPU_PLANES=$pl"; ( time BFC_GPU_PLANES=$pl oracle/_ref/bfc-dropin -E -s 250m -k 33 -t64 /dev/shm/c3e.fq ) 2>&1 | grep -E "^real|T::|Real time" | tail -16; echo; done
rm -f /dev/shm/c3e.fq
