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


def test_host_api_under_asan(tmp_path):
    exe = os.path.join(ROOT, "build", "asan_host")
    _build(exe, [os.path.join(ROOT, "scripts", "asan", "host_main.c"), os.path.join(ROOT, "bfc_amd", "csrc", "bfc_host.c")])
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "problems: 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
