"""Parse rate of the host ingest (no GPU): serial kseq-grammar parser vs the multi-threaded fast path, on the c2 FASTQ in tmpfs."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bfc_amd import _lib, gen
fq = "/dev/shm/c2_ingest.fq"
rs = gen.ReadSet(seed=2, G=4_600_000, cov=float(sys.argv[1]) if len(sys.argv) > 1 else 100)
if not os.path.exists(fq):
    rs.fastq(fq)
size = os.path.getsize(fq)
os.environ["BFC_INGEST_NOHASH"] = "1"
L = _lib.load()
out = (C.c_uint64 * 7)()
for threads in [0, 1, 2, 4, 8, 16, 32, 64]:
    if threads > (os.cpu_count() or 1):
        break
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        L.bfc_ingest_digest(fq.encode(), 100000000, 110000000, threads, out)
        best = min(best, time.perf_counter() - t0)
    print("threads %2d (%s): %.3f s = %.2f GB/s of FASTQ, %d batches (%d fast), %d reads" % (threads, "serial parser" if threads == 0 else "fast path", best, size / best / 1e9, out[0], out[6], out[1]))
os.remove(fq)
