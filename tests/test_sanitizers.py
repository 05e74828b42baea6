"""Host code under AddressSanitizer + UBSan (CPU only): the ingest parsers over damaged inputs and the reference-shaped table / bloom
host API (scripts/asan/*).  The harnesses are compiled here with gcc; skipped when the sanitizer runtime is not installed."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["gcc", "-g", "-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "bfc_amd", "csrc")]


def _build(out, srcs):
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    r = subprocess.run(FLAGS + ["-o", out] + srcs + ["-lz", "-lpthread"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no sanitizer build here: " + r.stderr[-300:])


def test_ingest_under_asan():
    _build(os.path.join(ROOT, "build", "asan_ingest"), [os.path.join(ROOT, "scripts", "asan", "ingest_main.c")])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "asan", "run_ingest.py"), "9"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "problems: 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_parallel_inflate_under_asan():
    _build(os.path.join(ROOT, "build", "asan_pgz"), [os.path.join(ROOT, "scripts", "asan", "pgz_main.c")])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "asan", "run_pgz.py"), "26"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "problems: 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_threaded_ingest_under_tsan(tmp_path):
    """the ingest's worker pool, the chained walks and the parallel inflate under ThreadSanitizer: a plain and a gzip'ed FASTQ, small and large batches"""
    import gzip
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_ingest as T
    exe = os.path.join(ROOT, "build", "tsan_ingest")
    r = subprocess.run(["gcc", "-g", "-O1", "-fsanitize=thread", "-fno-omit-frame-pointer", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "bfc_amd", "csrc"),
                        "-o", exe, os.path.join(ROOT, "scripts", "asan", "ingest_main.c"), "-lz", "-lpthread"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no ThreadSanitizer build here: " + r.stderr[-300:])
    data = T._fastq(np.random.default_rng(1), 20000, 50, 150)
    fq, gz = str(tmp_path / "t.fq"), str(tmp_path / "t.fq.gz")
    open(fq, "wb").write(data); open(gz, "wb").write(gzip.compress(data, 6))
    env = dict(os.environ, BFC_INGEST_GZ_MIN="0", BFC_INGEST_GZ_CHUNK="30000", BFC_INGEST_MIN_SLICE="4096", TSAN_OPTIONS="halt_on_error=0")
    outs = set()
    for fn in (fq, gz):
        for chunk in ("200000", "100000000"):
            r = subprocess.run([exe, fn, chunk, str(1 << 24), "8"], capture_output=True, text=True, env=env, timeout=600)
            assert r.returncode == 0 and not r.stderr.strip(), (fn, chunk, r.stderr[:1500])
            outs.add((chunk, " ".join(r.stdout.split()[1:5])))
    assert len(outs) == 2  # the gzip'ed file parses into the plain file's batches
    # round 6: the same text through a FIFO -- the pipe's reader thread, the ring's head / tail hand-over and the walks on its windows under ThreadSanitizer
    fifo = str(tmp_path / "in.fifo")
    os.mkfifo(fifo)
    for chunk in ("200000", "100000000"):
        w = subprocess.Popen(["sh", "-c", "cat %s > %s" % (fq, fifo)])
        r = subprocess.run([exe, fifo, chunk, str(1 << 24), "8"], capture_output=True, text=True, env=dict(env, BFC_INGEST_RING=str(4 << 20)), timeout=600)
        w.wait()
        assert r.returncode == 0 and not r.stderr.strip(), (chunk, r.stderr[:1500])
        assert (chunk, " ".join(r.stdout.split()[1:5])) in outs


def test_host_api_under_asan(tmp_path):
    exe = os.path.join(ROOT, "build", "asan_host")
    _build(exe, [os.path.join(ROOT, "scripts", "asan", "host_main.c"), os.path.join(ROOT, "bfc_amd", "csrc", "bfc_host.c")])
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "problems: 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
