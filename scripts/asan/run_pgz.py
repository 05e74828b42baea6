"""Drive build/asan_pgz over intact and damaged gzip files: no sanitizer report; the text is zlib's or the file is refused."""
import os, subprocess, sys, tempfile, zlib, gzip
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_pgz as P
import test_ingest as T

exe = os.path.join(ROOT, "build", "asan_pgz")
n_seed = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
with tempfile.TemporaryDirectory() as d:
    fn = os.path.join(d, "x.gz")
    for seed in range(first, first + n_seed):
        rng = np.random.default_rng(seed)
        kind = sorted(P.FILES)[seed % len(P.FILES)]
        text = T._fastq(rng, int(rng.integers(200, 3000)), 20, 150) if seed % 5 else rng.integers(0, 256, 100_000).astype(np.uint8).tobytes()
        z = bytearray(P.FILES[kind](text))
        for rep in range(4):
            y = bytearray(z)
            if rep:  # damage: flipped bits, a block of noise, truncation, a piece cut out
                for _ in range(int(rng.integers(1, 4))):
                    op = int(rng.integers(0, 4))
                    if op == 0 and len(y) > 30:
                        y[int(rng.integers(0, len(y)))] ^= 1 << int(rng.integers(0, 8))
                    elif op == 1 and len(y) > 200:
                        p = int(rng.integers(0, len(y) - 64)); y[p:p + 64] = rng.integers(0, 256, 64).astype(np.uint8).tobytes()
                    elif op == 2 and len(y) > 40:
                        y = y[:int(rng.integers(18, len(y)))]
                    elif len(y) > 400:
                        p = int(rng.integers(20, len(y) - 100)); del y[p:p + int(rng.integers(1, 80))]
            if len(y) < 18 or y[:2] != b"\x1f\x8b":
                continue
            open(fn, "wb").write(bytes(y))
            good, ok = P._zlib_all(bytes(y))
            for threads, chunk, window in ((1, 1 << 20, 1 << 30), (5, 3000, 50_000), (8, 200, 1 << 30), (3, 64, 1000)):
                r = subprocess.run([exe, fn, str(threads), str(chunk), str(window)], capture_output=True, text=True, env=env)
                if r.returncode != 0 or r.stderr.strip():
                    bad += 1
                    print("seed", seed, rep, kind, threads, chunk, "rc", r.returncode, r.stderr[:1500])
                    open("/tmp/asan_pgz_fail_%d.gz" % bad, "wb").write(bytes(y))
                    continue
                f = r.stdout.split()
                if int(f[0]) == 0:
                    if not ok or (int(f[1]), int(f[2])) != (len(good), zlib.crc32(good)):
                        bad += 1; print("seed", seed, rep, kind, threads, chunk, "WRONG TEXT accepted", f, len(good), ok)
                        open("/tmp/asan_pgz_fail_%d.gz" % bad, "wb").write(bytes(y))
                elif ok:
                    bad += 1; print("seed", seed, rep, kind, threads, chunk, "intact file refused", f)
                    open("/tmp/asan_pgz_fail_%d.gz" % bad, "wb").write(bytes(y))
print("problems:", bad)
sys.exit(1 if bad else 0)
