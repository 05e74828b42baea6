"""Config c3 of BASELINE.json on one MI355X: human-chr1-sized genome (248 Mbp, bfcgen seed 3) at 30x, 150 bp reads, k=33.
Stages the whole read set in HBM, counts it with the given batch size(s), and checks size-independent properties:
  * the GPU's k-mer count equals the host's (sum over ACGT runs of len-k+1),
  * sum_i i*cnt[i] == n_seen when no counter saturated (a random genome has no repeats: every bfc_ch_insert call is one count),
  * batch-size invariance: bloom popcount / FNV-1a, n_seen, distinct keys and both histograms are equal for every batch size.
Not the headline bench (that is c2, bench.py); numbers quoted in DESIGN.md.

    python scripts/c3_run.py [--b 35] [--batch-reads 4194304,8388608] [--cov 30] [--G 248000000]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bfc_amd
import oracle
from bfc_amd import gen

ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, default=33)
ap.add_argument("--b", type=int, default=35)
ap.add_argument("--G", type=int, default=248_000_000)
ap.add_argument("--cov", type=float, default=30.0)
ap.add_argument("--seed", type=int, default=3)
ap.add_argument("--batch-reads", default="4194304,8388608")
ap.add_argument("--filter-mode", type=int, default=0, help="1: `bfc -1` count pass (two bloom filters, no table)")
ap.add_argument("--digest", type=int, default=1, help="bring bloom + table to the host and compare across batch sizes")
args = ap.parse_args()
K = args.k
t0 = time.time()
rs = gen.ReadSet(seed=args.seed, G=args.G, cov=args.cov)
stride = rs.L + 1
n_reads = rs.n_reads
print("[c3] %d reads of %d bp, genome %d bp (%.1fs)" % (n_reads, rs.L, args.G, time.time() - t0), flush=True)
sizes = [int(v) for v in args.batch_reads.split(",")]
g = bfc_amd.GpuCounter(K, args.b, filter_mode=args.filter_mode, max_batch_pos=max(sizes) * stride)
d_seq = g.dev_alloc(n_reads * stride); d_qual = g.dev_alloc(n_reads * stride)
bad_tab = np.ones(256, dtype=bool); bad_tab[np.frombuffer(b"ACGTacgt", dtype=np.uint8)] = False
n_kmers = 0
CH = 2_000_000
for r0 in range(0, n_reads, CH):  # generate, count k-mers on the host, upload
    r1 = min(n_reads, r0 + CH)
    seq, qual, off = rs.reads(r0, r1)
    s = seq.reshape(r1 - r0, rs.L)
    bad = bad_tab[s]
    n_kmers += (r1 - r0) * (rs.L - K + 1)
    for r in np.nonzero(bad.any(axis=1))[0]:
        run = 0; c = 0
        for v in bad[r]:
            run = 0 if v else run + 1
            c += run >= K
        n_kmers += c - (rs.L - K + 1)
    g.h2d(d_seq + r0 * stride, bfc_amd.to_stream(seq, off)); g.h2d(d_qual + r0 * stride, bfc_amd.to_stream(qual, off))
    del seq, qual, s, bad
print("[c3] %d k-mers; read set staged in HBM (%.1fs)" % (n_kmers, time.time() - t0), flush=True)

results = []
for br in sizes:
    g.reset(); g.sync(); g.stage_ms(reset=True)
    t1 = time.perf_counter()
    for r0 in range(0, n_reads, br):
        r1 = min(n_reads, r0 + br)
        g.count_dev(d_seq + r0 * stride, d_qual + r0 * stride, (r1 - r0) * stride)
    g.sync()
    dt = time.perf_counter() - t1
    st = g.stats()
    ms, nb = g.stage_ms()
    assert st["n_kmers"] == n_kmers, (st["n_kmers"], n_kmers)
    res = dict(batch_reads=br, batches=nb, wall_s=round(dt, 4), G_kmers_per_s=round(n_kmers / dt / 1e9, 3), n_kmers=n_kmers, n_seen=st["n_seen"], n_keys=st["n_keys"],
               slow_buckets=st["slow_buckets"], tab_cshift=st["tab_cshift"], table=g.table_info(), stage_ms={k_: round(v, 2) for k_, v in ms.items()},
               bloom_GBps_algorithmic=round(128 * n_kmers / (ms["bloom"] * 1e-3) / 1e9, 1), bloom_frac=round(128 * n_kmers / (ms["bloom"] * 1e-3) / 1e9 / 8000, 4))
    if args.digest:
        t2 = time.time()
        bits = g.bloom_bytes()
        res["bloom_popcount"] = int(oracle.lib().orc_popcount_bytes(bits.ctypes.data, len(bits)))
        res["bloom_fnv1a64"] = "%016x" % int(oracle.lib().orc_fnv1a64(bits.ctypes.data, len(bits)))
        del bits
        if args.filter_mode:
            bits = g.bloom_bytes(1)
            res["bloom_hi_popcount"] = int(oracle.lib().orc_popcount_bytes(bits.ctypes.data, len(bits)))
            res["hist_bytes"] = "%016x" % int(oracle.lib().orc_fnv1a64(bits.ctypes.data, len(bits)))
            del bits
        else:
            t = g.export_table()
            mode, cnt, high = t.hist()
            assert t.count() == st["n_keys"]
            res.update(hist_mode=int(mode), cnt_sat=int(cnt[255]), sum_i_cnt=int((np.arange(256, dtype=np.uint64) * cnt).sum()),
                       cnt_head=[int(v) for v in cnt[1:6]], high_head=[int(v) for v in high[0:4]], hist_digest="%016x" % (hash((cnt.tobytes(), high.tobytes())) & (2**64 - 1)))
            res["hist_bytes"] = (cnt.tobytes() + high.tobytes()).hex()
            if res["cnt_sat"] == 0:
                assert res["sum_i_cnt"] == st["n_seen"], (res["sum_i_cnt"], st["n_seen"])
            t.close()
        res["digest_s"] = round(time.time() - t2, 1)
    results.append(res)
    print(json.dumps({k_: v for k_, v in res.items() if k_ != "hist_bytes"}), flush=True)
if args.digest and len(results) > 1:
    for r in results[1:]:
        for key in ("n_seen", "n_keys", "bloom_popcount", "bloom_fnv1a64", "hist_bytes"):
            assert r[key] == results[0][key], "batch size changes %s" % key
    print("[c3] batch-size invariance holds for", [r["batch_reads"] for r in results], flush=True)
g.dev_free(d_seq); g.dev_free(d_qual); g.close()
