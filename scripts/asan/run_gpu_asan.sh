#!/bin/bash
# On the GPU box: the ASan drop-in (scripts/asan/build_gpu_asan.sh) against the plain drop-in on the same inputs.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
{
python - <<'PY'
import sys; sys.path.insert(0,'.')
from bfc_amd import gen
rs = gen.ReadSet(seed=7, G=400_000, cov=60)
rs.fastq('/dev/shm/a.fq'); print('reads', rs.n_reads)
data = open('/dev/shm/a.fq','rb').read()
lines = data.split(b'\n')
# damaged copy: a truncated quality line, a doubled header, junk, CRLF, no final newline
lines[4*1000+3] = lines[4*1000+3][:50]; lines.insert(4*2000, lines[4*2000]); lines[4*3000+2+1] += b'\r'; lines.insert(4*5000+1, b'@junk+>')
open('/dev/shm/b.fq','wb').write(b'\n'.join(lines)[:-7])
import gzip; open('/dev/shm/a.fq.gz','wb').write(gzip.compress(data, 1))
PY
# leak checking off here: LeakSanitizer's report _exit()s from an atexit handler, before stdio's final flush, and libhsa's own leaks would cut stdout short
export ASAN_OPTIONS=protect_shadow_gap=0:detect_leaks=0:abort_on_error=0:log_path=/dev/shm/asan_log
A=oracle/_ref/bfc-dropin-asan; P=oracle/_ref/bfc-dropin-gputrim
run() { # name, args...: both binaries, compare stdout and dump
  n=$1; shift
  timeout 600 $A "$@" -d /dev/shm/$n.a.hash > /dev/shm/$n.a.out 2> /dev/shm/$n.a.err; ra=$?
  timeout 600 $P "$@" -d /dev/shm/$n.p.hash > /dev/shm/$n.p.out 2> /dev/shm/$n.p.err; rp=$?
  s=same; cmp -s /dev/shm/$n.a.out /dev/shm/$n.p.out || s=STDOUT-DIFFERS
  if [ -f /dev/shm/$n.p.hash ]; then cmp -s /dev/shm/$n.a.hash /dev/shm/$n.p.hash || s="$s DUMP-DIFFERS"; fi
  echo "== $n: asan rc=$ra plain rc=$rp $s ($(wc -c < /dev/shm/$n.a.out) bytes of stdout)"
  grep -E "ERROR|E::" /dev/shm/$n.a.err | head -5
  rm -f /dev/shm/$n.*
}
export BFC_GPU_EXACT_DUMP=1
run count_t1   -E -k31 -t1 /dev/shm/a.fq
run count_t8   -E -k31 -t8 /dev/shm/a.fq
run count_gz   -E -k33 -t4 /dev/shm/a.fq.gz
run count_bad  -E -k31 -t4 /dev/shm/b.fq
run count_bad1 -E -k31 -t1 /dev/shm/b.fq
BFC_GPU_BATCH=3000000 run count_small_batches -E -k31 -t4 /dev/shm/a.fq
run count_k51  -E -k51 -t4 /dev/shm/a.fq
run correct    -k31 -t4 /dev/shm/a.fq
unset BFC_GPU_EXACT_DUMP
runtrim() { n=$1; shift
  timeout 600 $A "$@" > /dev/shm/$n.a.out 2> /dev/shm/$n.a.err; ra=$?
  timeout 600 $P "$@" > /dev/shm/$n.p.out 2> /dev/shm/$n.p.err; rp=$?
  s=same; cmp -s /dev/shm/$n.a.out /dev/shm/$n.p.out || s=STDOUT-DIFFERS
  echo "== $n: asan rc=$ra plain rc=$rp $s ($(wc -c < /dev/shm/$n.a.out) bytes of stdout)"; grep -E "ERROR|E::" /dev/shm/$n.a.err | head -5; rm -f /dev/shm/$n.*; }
runtrim trim     -1 -k51 -b30 -t4 /dev/shm/a.fq
runtrim trim_bad -1 -k31 -b30 -t1 /dev/shm/b.fq
runtrim trim_gz  -1 -k51 -s 1m -t4 /dev/shm/a.fq.gz
ASAN_OPTIONS=protect_shadow_gap=0:detect_leaks=1:log_path=/dev/shm/leak_log $A -E -k31 -t4 /dev/shm/a.fq > /dev/null 2>&1
ASAN_OPTIONS=protect_shadow_gap=0:detect_leaks=1:log_path=/dev/shm/leak_log $A -1 -k51 -b30 -t4 /dev/shm/a.fq > /dev/null 2>&1
echo "== leak reports with a frame in this binary (the HSA runtime's own are ignored): $(cat /dev/shm/leak_log* 2>/dev/null | grep -c dropin-asan)"
cat /dev/shm/leak_log* 2>/dev/null | grep -B8 dropin-asan | head -60; rm -f /dev/shm/leak_log*
echo "== sanitizer logs:"; ls /dev/shm/asan_log* 2>/dev/null | head; for f in /dev/shm/asan_log*; do [ -f "$f" ] && { echo "--- $f"; head -60 "$f"; }; done
rm -f /dev/shm/a.fq /dev/shm/b.fq /dev/shm/a.fq.gz /dev/shm/asan_log*
} > gpurun_out/asan_gpu.log 2>&1; tail -c 6000 gpurun_out/asan_gpu.log
