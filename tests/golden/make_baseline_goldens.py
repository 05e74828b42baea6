#!/usr/bin/env python3
"""Known answers of THE REFERENCE ITSELF at BASELINE.json's single-GPU shapes (c2, c3) and at c5's parameters.

    make -C oracle && python tests/golden/make_baseline_goldens.py [c2 c3 c5s ...]

Runs the reference's own functions in file order (`bfc -t1` semantics: oracle/_ref/libbfcref.so = /root/reference
compiled in place, harness oracle/ref_shim.c, count.c:72-89 / count.c:127-157) over the full synthetic read set of
each configuration and writes tests/golden/baseline.json: k-mer / high / seen totals, distinct keys, both histograms,
bloom popcount + FNV-1a, the layout-free L1 digest of the table.  Data only.  Build container only
(c3 takes ~25 min of one core and ~12 GB; the GPU box never runs this).

bench.py compares its headline run with the c3 entry ("verified": true); tests/test_gpu_baseline_shapes.py compares the
GPU path with every entry.
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from bfc_amd import gen  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("BASELINE_OUT") or os.path.join(HERE, "baseline.json")

# name: generator arguments, k, bf_shift (as `-s` / the default give), filter_mode
CASES = {
    # c1 (BASELINE.json configs[0]): the reference's own CPU-runnable plumbing case, `bfc -t1 -E -k31` with the default -b33; its entry also holds
    # the md5 of the reference BINARY's `-d` dump of the FASTQ file bfcgen writes for it (count.c:127-157, htab.c:129-149)
    "c1": dict(gen=dict(seed=1, G=4_600_000, cov=1.0), k=31, b=33, fm=0, binary_dump=True),
    "c2": dict(gen=dict(seed=2, G=4_600_000, cov=100.0), k=31, b=33, fm=0),
    "c3": dict(gen=dict(seed=3, G=248_000_000, cov=30.0), k=33, b=35, fm=0),     # `-s 250m -k33` => -b35
    # c5's parameters (`-s 3g -k51 -1`: -b37, two 16 GiB filters, 20-byte records, 10+10 scatter levels) on a read set a CPU finishes
    "c5s": dict(gen=dict(seed=5, G=20_000_000, cov=30.0), k=51, b=37, fm=1),
    # c4's parameters (`-s 3g`: k=33, -b37, table mode) on the same small read set
    "c4s": dict(gen=dict(seed=5, G=20_000_000, cov=30.0), k=33, b=37, fm=0),
    # an EIGHTH of c4 itself (round 4): `-s 3g` => k=33, -b37 on a 387.5 Mbp genome (bfcgen seed 4) at 30x -- 77.5 M reads, 9.1 G k-mers, 2^20 regions
    # loaded as a whole GPU's share of the 8-GPU configuration would be; ~2.5 h of one core and ~35 GB here.  bench.py's secondary `c4e` and
    # tests/test_gpu_baseline_shapes.py hold the GPU path to it
    "c4e": dict(gen=dict(seed=4, G=387_500_000, cov=30.0), k=33, b=37, fm=0),
    # an EIGHTH of c5 (round 5): the same 77.5 M reads at c5's parameters, `-s 3g -k51 -1` => -b37, two 16 GiB filters (count.c:67-68), 16-byte records;
    # 7.7 G k-mers.  ~2 h of one core and ~40 GB here.  bench.py's secondary `c5e` and tests/test_gpu_baseline_shapes.py hold the GPU path to it
    "c5e": dict(gen=dict(seed=4, G=387_500_000, cov=30.0), k=51, b=37, fm=1),
    # round 6 -- the two below are made by the block-partitioned harness (oracle/ref_shim.c: ref_count_batch_blocks, MT threads; the same answers as
    # the sequential one: tests/test_oracle.py::test_block_partitioned_harness_equals_the_sequential_one, and `same_as` re-derives a committed
    # sequential entry at full size before anything is added to it)
    # config c5's QUERY pass: c5e's count again (must reproduce the committed c5e entry), then the reference's own trim pass (worker_ec ->
    # max_streak + keep rule, correct.c:478-497,557-567, via oracle/ref_shim_ec.c) over the same 77.5 M reads against the reference's bf_high
    # round 6: c4e again through the block-partitioned harness -- must reproduce the committed sequential entry, to which the filter's parallel digest (bf_mix64)
    # is added so that the GPU suite need not take FNV-1a of 16 GiB (17 s of one core) for every shape test
    "c4e_mix": dict(gen=dict(seed=4, G=387_500_000, cov=30.0), k=33, b=37, fm=0, mt=True, same_as="c4e", merge_into="c4e"),
    "c5e_trim": dict(gen=dict(seed=4, G=387_500_000, cov=30.0), k=51, b=37, fm=1, mt=True, same_as="c5e", trim=True),
    # the reference's one published command line, `bfc -s 3g -k55` (tex/README.md:26): c4e's reads at k=55, -b37, TABLE mode -- 2^24 sub-tables
    # (htab.c:19-34), the lossy key of k >= 38 (htab.c:45-58), 20-byte records on the GPU
    "c4e_k55": dict(gen=dict(seed=4, G=387_500_000, cov=30.0), k=55, b=37, fm=0, mt=True),
}
MT = int(os.environ.get("GOLDEN_THREADS", "6"))


def run(name):
    cs = CASES[name]
    t0 = time.time()
    rs = gen.ReadSet(**cs["gen"])
    c = oracle.Counter(cs["k"], cs["b"], filter_mode=cs["fm"], impl="ref")
    CH = 2_000_000
    for r0 in range(0, rs.n_reads, CH):
        seq, qual, off = rs.reads(r0, min(rs.n_reads, r0 + CH))
        if cs.get("mt"):
            c.count_blocks(seq, qual, off, MT)
        else:
            c.count(seq, qual, off)
        print("[%s] %d / %d reads  %.0fs" % (name, min(rs.n_reads, r0 + CH), rs.n_reads, time.time() - t0), file=sys.stderr, flush=True)
    st = c.stats()
    pop, fnv = c.bloom_checksums()
    mix = gen.bitmap_mix64(c.bloom_view())
    e = dict(name=name, gen=cs["gen"], k=cs["k"], b=cs["b"], filter_mode=cs["fm"], n_reads=rs.n_reads,
             n_kmers=st["n_kmers"], n_high=st["n_high"], n_seen=st["n_seen"], hash_xor="%016x" % st["hash_xor"],
             bf_popcount=pop, bf_fnv1a64="%016x" % fnv, bf_mix64="%016x" % mix)
    if cs.get("mt"):
        e["harness"] = "ref_count_batch_blocks, %d threads" % MT
    if cs["fm"]:
        pop2, fnv2 = c.bloom_checksums(high=True)
        e.update(bf_high_popcount=pop2, bf_high_fnv1a64="%016x" % fnv2, bf_high_mix64="%016x" % gen.bitmap_mix64(c.bloom_view(True)))
    else:
        mode, cnt, high = c.table_hist()
        e.update(distinct=c.table_count(), hist_mode=int(mode), cnt=[int(v) for v in cnt], high=[int(v) for v in high])
        with tempfile.NamedTemporaryFile(suffix=".hash", dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tf:
            c.dump(tf.name)
            kk, l_pre, sizes, slots = oracle.parse_dump(tf.name)
        e.update(l_pre=l_pre, l1_digest=oracle.l1_digest(sizes, slots))
    if cs.get("same_as"):  # the block-partitioned harness at full size against the committed SEQUENTIAL entry of the same configuration
        old = {x["name"]: x for x in json.load(open(OUT))}[cs["same_as"]]
        for f in ("n_reads", "n_kmers", "n_high", "n_seen", "hash_xor", "bf_popcount", "bf_fnv1a64", "bf_high_popcount", "bf_high_fnv1a64", "distinct", "hist_mode", "cnt", "high", "l1_digest"):
            if f in old:
                assert e[f] == old[f], (f, e[f], old[f])
        e["reproduces"] = cs["same_as"]
        print("[%s] reproduces the sequential entry %s  %.0fs" % (name, cs["same_as"], time.time() - t0), file=sys.stderr, flush=True)
    if cs.get("trim"):
        # (start, end) per read as the reference's worker_ec leaves them, -1 / -1 for a dropped read; FNV-1a/64 over the int32 pairs in read order
        kept = bases = 0
        h = 0xcbf29ce484222325
        G = gen._L()
        for r0 in range(0, rs.n_reads, CH):
            seq, qual, off = rs.reads(r0, min(rs.n_reads, r0 + CH))
            st_, en_ = oracle.ref_trim(c._bf(True), cs["k"], seq, off, 0.9, MT)
            m = st_ >= 0
            kept += int(m.sum()); bases += int((en_[m] - st_[m]).sum())
            pairs = np.ascontiguousarray(np.stack([st_, en_], axis=1), dtype="<i4")
            h = int(G.bfcgen_fnv1a64_from(h, pairs.ctypes.data, pairs.nbytes))
            print("[%s] trim %d / %d reads  %.0fs" % (name, min(rs.n_reads, r0 + CH), rs.n_reads, time.time() - t0), file=sys.stderr, flush=True)
        e.update(min_frac=0.9, trim_reads_kept=kept, trim_bases_kept=bases, trim_windows_fnv1a64="%016x" % h, trim_queries=e["n_kmers"])
    c.close()
    if cs.get("binary_dump"):  # the unmodified reference binary on the file itself: `bfc-ref -t1 -E -k K -b B -d dump reads.fq`
        import subprocess
        d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        fq, dump = os.path.join(d, name + ".fq"), os.path.join(d, name + ".hash")
        rs.fastq(fq)
        r = subprocess.run([os.path.join(oracle.REF_DIR, "bfc-ref"), "-t", "1", "-E", "-k", str(cs["k"]), "-b", str(cs["b"]), "-d", dump, fq], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-500:]
        e.update(fastq_md5=oracle.md5_file(fq), ref_dump_md5=oracle.md5_file(dump), ref_cmd="bfc-ref -t 1 -E -k %d -b %d -d dump %s.fq" % (cs["k"], cs["b"], name))
        k2, l2, sz2, sl2 = oracle.parse_dump(dump)
        assert oracle.l1_digest(sz2, sl2) == e["l1_digest"], "harness and binary disagree"  # harness == binary
        for f in (fq, dump):
            os.unlink(f)
        os.rmdir(d)
    e["ref_seconds"] = round(time.time() - t0, 1)
    print(e, file=sys.stderr, flush=True)
    return e


if __name__ == "__main__":
    assert oracle.have_ref(), "build oracle/_ref first (make -C oracle)"
    names = sys.argv[1:] or list(CASES)
    cur = {}
    if os.path.exists(OUT):
        cur = {e["name"]: e for e in json.load(open(OUT))}
    for n in names:
        e = run(n)
        tgt = CASES[n].get("merge_into")
        if tgt:  # a re-derivation that only ADDS digests to a committed entry it has just reproduced
            cur[tgt].update({k: e[k] for k in ("bf_mix64", "bf_high_mix64") if k in e})
            cur[tgt]["mix64_from"] = "%s: %s, reproducing every field of this entry" % (n, e["harness"])
        else:
            cur[n] = e
        json.dump([cur[k] for k in sorted(cur)], open(OUT, "w"), indent=1)
    print("wrote", OUT, sorted(cur))
