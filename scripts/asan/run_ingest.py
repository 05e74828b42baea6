"""Drive build/asan_ingest over the damaged inputs of tests/test_ingest.py: no sanitizer report, and the parsers agree."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_ingest as T

exe = os.path.join(ROOT, "build", "asan_ingest")
n_seed = int(sys.argv[1]) if len(sys.argv) > 1 else 60
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
with tempfile.TemporaryDirectory() as d:
    fn = os.path.join(d, "d.fq")
    for seed in range(first, first + n_seed):
        rng = np.random.default_rng(seed)
        kind = seed % 3
        if kind == 0:
            data = T._fastq(rng, int(rng.integers(1, 400)), 1, 120, crlf=rng.random() < 0.2)
        elif kind == 1:
            data = b"".join(b">f%d x\n" % r + rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), int(rng.integers(0, 200))).tobytes() + b"\n" for r in range(int(rng.integers(1, 300))))
        else:
            data = T._fastq(rng, 150, 10, 80) + b">fa\nACGTTGCA\nAC\n" + T._fastq(rng, 150, 10, 80)
        for rep in range(3):
            open(fn, "wb").write(T._mutate(rng, data))
            for chunk, cap in ((300, 1 << 22), (1 << 30, 1 << 22), (1 << 30, 20000)):
                outs = []
                for threads, ms in ((0, None), (3, None), (5, "64")):
                    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
                    if ms: env["BFC_INGEST_MIN_SLICE"] = ms
                    r = subprocess.run([exe, fn, str(chunk), str(cap), str(threads)], capture_output=True, text=True, env=env)
                    if r.returncode != 0 or r.stderr.strip():
                        bad += 1
                        print("seed", seed, rep, chunk, cap, threads, "rc", r.returncode, r.stderr[:1500])
                        open("/tmp/asan_fail_%d.fq" % bad, "wb").write(open(fn, "rb").read())
                    outs.append(r.stdout)
                if len(set(outs)) != 1:
                    bad += 1; print("seed", seed, rep, chunk, cap, "parsers disagree", outs)
                # round 6: the same text through a FIFO (reader thread + ring mapped twice, the chained walks on windows of it, the serial parser FROM the ring) ...
                if cap > 20000:
                    fifo = os.path.join(d, "in.fifo")
                    if not os.path.exists(fifo):
                        os.mkfifo(fifo)
                    for threads, ms in ((3, None), (5, "64")):
                        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
                        if ms: env["BFC_INGEST_MIN_SLICE"] = ms
                        w = subprocess.Popen(["sh", "-c", "cat %s > %s" % (fn, fifo)])
                        r = subprocess.run([exe, fifo, str(chunk), str(cap), str(threads)], capture_output=True, text=True, env=env)
                        w.wait()
                        if r.returncode != 0 or r.stderr.strip() or r.stdout != outs[0]:
                            bad += 1; print("seed", seed, rep, chunk, cap, threads, "FIFO: rc", r.returncode, r.stdout, outs[0], r.stderr[:1500])
                    # ... and the batches as bit planes (AVX2 packing with masked tails against the per-position rule on the serial parser's byte streams)
                    pouts = []
                    for threads, ms in ((0, None), (3, None), (5, "64")):
                        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", ASAN_INGEST_PLANES="1")
                        if ms: env["BFC_INGEST_MIN_SLICE"] = ms
                        r = subprocess.run([exe, fn, str(chunk), str(cap), str(threads)], capture_output=True, text=True, env=env)
                        if r.returncode != 0 or r.stderr.strip():
                            bad += 1; print("seed", seed, rep, chunk, cap, threads, "planes: rc", r.returncode, r.stderr[:1500])
                        pouts.append(r.stdout)
                    if len(set(pouts)) != 1:
                        bad += 1; print("seed", seed, rep, chunk, cap, "planes differ", pouts)
            r = subprocess.run([exe, fn, "0", "0", "0", "hdr"], capture_output=True, text=True, env=dict(os.environ, UBSAN_OPTIONS="print_stacktrace=1"))
            if r.returncode != 0 or r.stderr.strip():
                bad += 1; print("seed", seed, rep, "keep_hdr parse: rc", r.returncode, r.stderr[:1500])
            if bad > 5: sys.exit(1)
print("done, problems:", bad)
sys.exit(1 if bad else 0)
