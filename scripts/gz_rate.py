"""Rate of gzip ingest (no GPU): zlib's one stream vs the parallel inflate (bfc_pgz.h) vs the whole ingest (inflate + fast-path parser),
on a gzip'ed c2-like FASTQ in tmpfs.  usage: gz_rate.py [coverage=25] [gzip level=6] [members=1]"""
import ctypes as C, os, subprocess, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bfc_amd import _lib, gen
cov = float(sys.argv[1]) if len(sys.argv) > 1 else 25
lvl = int(sys.argv[2]) if len(sys.argv) > 2 else 6
fq = "/dev/shm/gz_rate.fq"
gen.ReadSet(seed=2, G=4_600_000, cov=cov).fastq(fq)
size = os.path.getsize(fq)
t0 = time.perf_counter(); subprocess.run("gzip -%d -k -f %s" % (lvl, fq), shell=True, check=True); t_gz = time.perf_counter() - t0
gz = fq + ".gz"
rep_n = int(sys.argv[3]) if len(sys.argv) > 3 else 1   # the same member several times over: a longer stream without the wait for gzip
if rep_n > 1:
    z = open(gz, "rb").read()
    with open(gz, "wb") as f:
        for _ in range(rep_n):
            f.write(z)
    size *= rep_n
print("FASTQ %.3f GB, gzip -%d -> %.3f GB in %.1f s; %d host threads" % (size / 1e9, lvl, os.path.getsize(gz) / 1e9, t_gz, os.cpu_count()))
t0 = time.perf_counter(); subprocess.run("gzip -dc %s > /dev/null" % gz, shell=True, check=True); t = time.perf_counter() - t0
print("gzip -dc: %.2f s = %.2f GB/s of text" % (t, size / t / 1e9))
L = _lib.load()
os.environ["BFC_INGEST_NOHASH"] = "1"
out5 = (C.c_uint64 * 5)(); out7 = (C.c_uint64 * 7)()
t0 = time.perf_counter(); L.bfc_ingest_digest(gz.encode(), 100000000, 110000000, 0, out7); t = time.perf_counter() - t0
print("ingest, serial (gzread + kseq-grammar parser): %.2f s = %.2f GB/s of text, %d reads" % (t, size / t / 1e9, out7[1]))
n_reads = out7[1]
for threads in [1, 4, 16, 32, 64]:
    if threads > (os.cpu_count() or 1):
        break
    best = 1e9
    for rep in range(2):
        t0 = time.perf_counter(); rc = L.bfc_pgz_digest(gz.encode(), threads, 2 << 20, 1 << 30, out5); best = min(best, time.perf_counter() - t0)
    assert rc == 0 and out5[0] == size
    b2 = 1e9
    for rep in range(2):
        t0 = time.perf_counter(); L.bfc_ingest_digest(gz.encode(), 100000000, 110000000, threads, out7); b2 = min(b2, time.perf_counter() - t0)
    assert out7[1] == n_reads and out7[6] == out7[0]
    print("threads %2d: inflate alone %.2f s = %.2f GB/s; ingest (inflate + parse into batches) %.2f s = %.2f GB/s of text; %d guessed / %d redone pieces"
          % (threads, best, size / best / 1e9, b2, size / b2 / 1e9, out5[2], out5[3]))
os.remove(fq); os.remove(gz)
