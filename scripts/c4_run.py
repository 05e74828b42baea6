"""Configs c4 / c5 of BASELINE.json at full size on ONE MI355X: human-sized genome (3.1 Gbp, bfcgen seed 4) at 30x, 150 bp reads,
`-s 3g` => k=33, -b37 (16 GiB filter).  190 GB of reads cannot be resident: batches are generated on the host (a thread one batch ahead)
and submitted with bfcg_count_batch_host, so the wall time is the generator's; the GPU time is the sum of the batches' stage times.
Checks: the GPU's k-mer count equals the host's formula; c4: table statistics; c5 (--filter-mode 1, k=51): both filters' popcounts.

    python scripts/c4_run.py [--cov 30] [--batch-reads 4194304] [--filter-mode 0|1] [--k 33]
"""
import argparse, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bfc_amd
from bfc_amd import gen

ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, default=33)
ap.add_argument("--b", type=int, default=37)
ap.add_argument("--G", type=int, default=3_100_000_000)
ap.add_argument("--cov", type=float, default=30.0)
ap.add_argument("--seed", type=int, default=4)
ap.add_argument("--batch-reads", type=int, default=4194304)
ap.add_argument("--filter-mode", type=int, default=0)
ap.add_argument("--trim", type=int, default=0, help="1 (with --filter-mode 1): then the trim pass of bfc -1 over the same reads (bloom query kernel + longest streak)")
ap.add_argument("--export", type=int, default=0, help="1 (table mode): bring the count table to the host as bfc_count does and check it there (bfc_ch_count, bfc_ch_hist)")
ap.add_argument("--per-batch", type=int, default=0, help="1: stage times of the last finalised library batch after every call")
ap.add_argument("--popcount", type=int, default=0, help="1: bring the filter(s) to the host and count their bits")
args = ap.parse_args()
K = args.k
t0 = time.time()
rs = gen.ReadSet(seed=args.seed, G=args.G, cov=args.cov)
stride, n_reads, br = rs.L + 1, rs.n_reads, args.batch_reads
print("[c4] %d reads of %d bp, genome %d bp (%.1fs)" % (n_reads, rs.L, args.G, time.time() - t0), flush=True)
g = bfc_amd.GpuCounter(K, args.b, filter_mode=args.filter_mode, max_batch_pos=br * stride)
print("[c4] context ready (%.1fs)" % (time.time() - t0), flush=True)
bad_tab = np.ones(256, dtype=bool); bad_tab[np.frombuffer(b"ACGTacgt", dtype=np.uint8)] = False


def make(r0):
    r1 = min(n_reads, r0 + br)
    seq, qual, off = rs.reads(r0, r1)
    s = seq.reshape(r1 - r0, rs.L)
    nk = (r1 - r0) * (rs.L - K + 1)
    bad = bad_tab[s]
    for r in np.nonzero(bad.any(axis=1))[0]:  # ~1 % of the reads carry an N
        run = 0; c = 0
        for v in bad[r]:
            run = 0 if v else run + 1
            c += run >= K
        nk += c - (rs.L - K + 1)
    return bfc_amd.to_stream(seq, off), bfc_amd.to_stream(qual, off), nk


nxt = {}
def prefetch(r0):
    nxt["v"] = make(r0)

n_kmers = 0; gpu_ms = dict(hist1=0.0, scatter1=0.0, level2=0.0, bloom=0.0, commit=0.0, total=0.0)
cur = make(0)
t1 = time.perf_counter()
nb = 0
for r0 in range(0, n_reads, br):
    th = None
    if r0 + br < n_reads:
        th = threading.Thread(target=prefetch, args=(r0 + br,)); th.start()
    s, q, nk = cur
    g.count_host(s, q)
    n_kmers += nk; nb += 1
    if args.per_batch:
        print("[c4] call %d: last batch %s %s" % (nb, {k_: round(v, 1) for k_, v in g.last_batch_ms().items()}, g.partition_info()), flush=True)
    if th:
        th.join(); cur = nxt["v"]
    if nb % 16 == 0:
        st = g.stats()
        print("[c4] %d batches, %.1f G k-mers, %.0f s wall; seen %d keys %d table 2^%d slots" % (nb, n_kmers / 1e9, time.perf_counter() - t1, st["n_seen"], st["n_keys"], 20 + st["tab_cshift"] if K <= 36 else 24 + st["tab_cshift"]), flush=True)
g.sync()
wall = time.perf_counter() - t1
st = g.stats()
ms, nbt = g.stage_ms()
assert st["n_kmers"] == n_kmers, (st["n_kmers"], n_kmers)
res = dict(config="c5 (bfc -1 count pass)" if args.filter_mode else "c4", k=K, b=args.b, reads=n_reads, batch_reads=br, batches=nbt, n_kmers=n_kmers, n_seen=st["n_seen"], n_keys=st["n_keys"],
           slow_buckets=st["slow_buckets"], partition=g.partition_info(), tab_cshift=st["tab_cshift"], wall_s_generator_bound=round(wall, 1), gpu_stage_ms={k_: round(v, 1) for k_, v in ms.items()},
           gpu_s=round(ms["total"] / 1e3, 2), G_kmers_per_gpu_s=round(n_kmers / ms["total"] / 1e6, 2), bloom_frac=round(128 * n_kmers / (ms["bloom"] * 1e-3) / 1e9 / 8000, 4))
if args.popcount:
    import oracle
    for which in ([0, 1] if args.filter_mode else [0]):
        bits = g.bloom_bytes(which)
        res["bloom%d_popcount" % which] = int(oracle.lib().orc_popcount_bytes(bits.ctypes.data, len(bits)))
        del bits
if args.export and not args.filter_mode:
    t2 = time.time(); tab = g.export_table(); t3 = time.time()
    mode, cnt, high = tab.hist(); t4 = time.time()
    assert tab.count() == st["n_keys"] == int(cnt.sum()) == int(high.sum()), (tab.count(), st["n_keys"], int(cnt.sum()))
    sat = int(cnt[255])
    lo = int((np.arange(256, dtype=np.uint64) * cnt).sum())  # = n_seen when no counter saturated, a lower bound otherwise
    assert lo <= st["n_seen"] and (sat > 0 or lo == st["n_seen"]), (lo, st["n_seen"], sat)
    res.update(export_s=round(t3 - t2, 2), export_GiB=round(8 * 2.0 ** ((20 if K <= 36 else 24) + st["tab_cshift"]) / 2 ** 30, 1), hist_s=round(t4 - t3, 2), hist_mode=int(mode), cnt_saturated=sat, sum_i_cnt=lo)
    tab.close()
if args.trim and args.filter_mode:
    t2 = time.time()
    bf = g.export_bloom(1, resident=True)  # as bfc_count does: host object + copy left in HBM for the trim context to adopt
    g.close(); g = None
    tr = bfc_amd.GpuTrimmer(K, bf, max_pos=br * stride, max_reads=br)
    res["trim_filter_adopted_from_hbm"] = tr.adopted
    print("[c4] second filter moved to the trimmer (%.1fs)" % (time.time() - t2), flush=True)
    kept = bases = 0; q_ms = 0.0; nq = 0
    cur = make(0)
    t3 = time.perf_counter()
    for r0 in range(0, n_reads, br):
        th = None
        if r0 + br < n_reads:
            th = threading.Thread(target=prefetch, args=(r0 + br,)); th.start()
        s, q, nk = cur
        n = len(s) // stride
        so = np.arange(n + 1, dtype=np.uint64) * np.uint64(stride)
        st_, en_ = tr.trim(s, so, 0.9)
        q_ms += tr.last_ms(); nq += nk
        k_ = st_ >= 0
        kept += int(k_.sum()); bases += int((en_[k_] - st_[k_]).sum())
        if th:
            th.join(); cur = nxt["v"]
    res.update(trim_reads_kept=kept, trim_bases_kept=bases, trim_queries=nq, trim_gpu_s=round(q_ms / 1e3, 2), trim_G_queries_per_gpu_s=round(nq / q_ms / 1e6, 2),
               trim_wall_s_generator_bound=round(time.perf_counter() - t3, 1))
    tr.close(); bf.close()
print(json.dumps(res), flush=True)
if g is not None:
    g.close()
