#!/usr/bin/env python3
"""bench.py -- headline benchmark of the k-mer counting hot path (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (config c2 of BASELINE.json, the configuration the metric is quoted on): synthetic
E. coli-sized genome (4.6 Mbp, seed 2) at 100x, 150 bp reads with 1 % substitutions, k=31, default
1 GiB bloom filter (-b 33, -H 4), l_pre 20 -- 3.07 M reads, 368 M k-mers per GPU.  A *step* is one
full count of that read set: reset of bloom filter + table, then every batch through
hash -> scatter -> bloom regions -> count table, inputs already resident in HBM.

One JSON line on stdout (rank 0).  `roofline` prices the bloom-region kernel (the kernel the
north star names) at SURVEY 8(d)'s algorithmic 128 B per k-mer against 8 TB/s, from HIP-event
time on the library's stream; `cpu_baseline` times the *reference binary* (oracle/_ref/bfc-ref,
built in place from /root/reference) on a bounded sample of the same reads on this box's host
cores.  The oracle is used nowhere in the timed path.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K, BF_SHIFT, N_HASHES, L_PRE, Q = 31, 33, 4, 20, 20
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BLOOM_BYTES_PER_KMER = 128   # one 64-byte block read + write per k-mer (bbf.c:31)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(rs, n_cores):
    """Reference `bfc -E -k31 -t<cores>` on a bounded read sample, net of its fixed setup."""
    ref = os.path.join(ROOT, "oracle", "_ref", "bfc-ref")
    if not os.path.exists(ref):
        return cpu_baseline_port(rs)
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    out = {}

    def run(fq, t):
        t0 = time.time()
        r = subprocess.run([ref, "-E", "-k", str(K), "-b", str(BF_SHIFT), "-t", str(t), fq], capture_output=True, text=True)
        m = re.search(r"Real time: ([0-9.]+) sec", r.stderr)
        return float(m.group(1)) if m else time.time() - t0

    one = os.path.join(shm, "bfc_bench_one.fq")
    rs.fastq(one, 0, 1)
    for label, t, n_reads in (("t1", 1, 60000), ("tN", n_cores, 1500000)):
        fq = os.path.join(shm, "bfc_bench_%s.fq" % label)
        n_reads = min(n_reads, rs.n_reads)
        rs.fastq(fq, 0, n_reads)
        seq, _, off = rs.reads(0, n_reads)
        import numpy as np
        kmers = count_kmers(seq, rs.L, K)
        setup = run(one, t)
        wall = run(fq, t)
        net = max(wall - setup, 1e-3)
        out[label] = dict(threads=t, reads=n_reads, kmers=int(kmers), wall_s=round(wall, 3), setup_s=round(setup, 3), mkmers_per_s=round(kmers / net / 1e6, 3))
        os.unlink(fq)
    os.unlink(one)
    best = max(out.values(), key=lambda d: d["mkmers_per_s"])
    return {"value": best["mkmers_per_s"], "unit": "M k-mers/s", "cores": best["threads"], "kind": "reference",
            "sample": "oracle/_ref/bfc-ref -E -k31 -b33 -t%d on the first %d reads of the same synthetic set (%d k-mers), "
                      "wall %.2fs minus %.2fs setup (1-read run)" % (best["threads"], best["reads"], best["kmers"], best["wall_s"], best["setup_s"]),
            "t1_mkmers_per_s": out["t1"]["mkmers_per_s"], "detail": out}


def cpu_baseline_port(rs):
    """Fallback when the reference binary did not travel: the single-threaded C restatement (oracle/) on a bounded sample."""
    import oracle
    n_reads = min(100000, rs.n_reads)
    seq, qual, off = rs.reads(0, n_reads)
    c = oracle.Counter(K, BF_SHIFT, q=Q, n_hashes=N_HASHES, l_pre=L_PRE)
    t0 = time.time()
    n = c.count(seq, qual, off)
    dt = time.time() - t0
    c.close()
    return {"value": round(n / dt / 1e6, 3), "unit": "M k-mers/s", "cores": 1, "kind": "port",
            "sample": "oracle/bfc_oracle.c (sequential restatement), first %d reads of the same synthetic set (%d k-mers) in %.2fs; "
                      "oracle/_ref/bfc-ref was not available on this box" % (n_reads, n, dt)}


def pmc_traffic():
    """HBM bytes per k_bloom launch measured with rocprofv3 PMC passes of this same command (committed summary)."""
    try:
        with open(os.path.join(ROOT, "profiles", "round1_pmc.json")) as f:
            return round(json.load(f)["k_bloom"]["hbm_bytes_per_launch"])
    except Exception:  # noqa: BLE001
        return None


def count_kmers(seq, L, k):
    """Number of bfc_kmer_insert calls: sum over ACGT runs of max(0, len-k+1) (vectorised, fixed-length reads)."""
    import numpy as np
    n = len(seq) // L
    s = seq.reshape(n, L)
    bad = ~np.isin(s, np.frombuffer(b"ACGTacgt", dtype=np.uint8))
    total = n * (L - k + 1)
    rows = np.nonzero(bad.any(axis=1))[0]
    for r in rows:  # ~1 % of reads carry an N
        run = 0; c = 0
        for v in bad[r]:
            run = 0 if v else run + 1
            c += run >= k
        total += c - (L - k + 1)
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-reads", type=int, default=int(os.environ.get("BFC_BENCH_BATCH_READS", 786432)))
    ap.add_argument("--cov", type=float, default=100.0, help="coverage of the 4.6 Mbp genome (100 = config c2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1)); local = int(os.environ.get("LOCAL_RANK", 0))
    # stdout carries exactly ONE line, the JSON: libraries that print there (RCCL's version banner at communicator creation) go to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    use_dist = world > 1 or bool(os.environ.get("BFC_BENCH_FORCE_DIST"))
    dist = None
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
        import torch  # first: libbfc_gpu.so then binds to the HIP runtime torch already loaded (one runtime per process)
        import torch.distributed as dist
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    import numpy as np
    import bfc_amd
    from bfc_amd import gen, build
    if rank == 0:
        build.build()
    if dist:
        dist.barrier()

    # ---- synthetic input (c2), one read set per rank (weak scaling: per-GPU work is fixed).  With N > 1 the N sets are one
    # data set in rank-major batch order: global batch t = rank 0's t-th slice, rank 1's t-th slice, ...
    t0 = time.time()
    rs = gen.ReadSet(seed=2 + rank, G=4_600_000, cov=args.cov, L=150, err=0.01)
    seq, qual, off = rs.reads()
    n_reads = rs.n_reads
    n_kmers = count_kmers(seq, rs.L, K)
    s_seq, s_qual = bfc_amd.to_stream(seq, off), bfc_amd.to_stream(qual, off)
    stride = rs.L + 1
    batch_reads = min(args.batch_reads, n_reads)

    # Weak scaling keeps the work PER GPU fixed: every rank brings its own c2 read set (another genome: seed 2 + rank) and owns 2^33 bits of
    # the filter, so with N ranks the job is one count of N x c2 into a filter of 2^33 x N bits (-b 33 + log2 N) -- what `bfc -s` does for an
    # N times larger genome.  With the filter fixed at -b33 every bloom region would receive N times a single-GPU batch's k-mers per global
    # batch, more than a region's LDS list takes: all of them on the exact but ~25x slower path (scripts/mg_load.py).
    def make_counter(n_ranks):
        shift = BF_SHIFT + (n_ranks.bit_length() - 1 if n_ranks > 1 else 0)
        return bfc_amd.GpuCounter(K, shift, q=Q, n_hashes=N_HASHES, l_pre=L_PRE, device=local, max_batch_pos=batch_reads * stride,
                                  rank=rank if n_ranks > 1 else 0, n_ranks=n_ranks)

    # owner-computes exchange over RCCL; a collective preflight decides for ALL ranks whether it is usable here
    mode = "1 GPU"
    eng = None
    if use_dist:
        from bfc_amd import dist as bdist
        import torch
        ok = 1
        try:
            g = make_counter(world)
            eng = bdist.GpuEngine(g)
            # a real (small) exchange through the very code the timed loop uses: 5 records per level-1 bucket from every rank
            nb1 = g.mg_info()["nb1"]
            eng.send[:5 * nb1 * eng.rec_words].fill_(rank + 1)
            seg = bdist.exchange(eng, np.full(nb1, 5, dtype=np.uint32))
            torch.cuda.synchronize()
            got = eng.recv[:5 * nb1 * eng.rec_words].view(world, -1)
            want = torch.arange(1, world + 1, device=got.device, dtype=got.dtype).view(world, 1).expand_as(got)
            if seg.shape != (world, nb1 // world) or int(seg.sum()) != 5 * nb1 or not bool((got == want).all()):
                raise RuntimeError("exchange preflight delivered wrong data")
        except Exception as e:  # noqa: BLE001
            log("[bench] rank %d: exchange path unavailable: %r" % (rank, e))
            ok = 0
        flag = torch.tensor([ok], device="cuda"); dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            mode = "dp%d: read shards per GPU (one c2 read set each), ONE filter of 2^%d bits (2^33 per GPU) and one table, bloom regions + table keys owned by one GPU each, 1 all-to-all of %d-byte k-mer records per batch (RCCL)" % (world, BF_SHIFT + world.bit_length() - 1, g.mg_info()["rec_bytes"])
        else:
            eng = None
            g = make_counter(1)
            mode = "%d independent read shards (exchange preflight failed: no cross-GPU counting)" % world
    else:
        g = make_counter(1)
    d_seq = g.dev_alloc(len(s_seq)); d_qual = g.dev_alloc(len(s_qual))
    g.h2d(d_seq, s_seq); g.h2d(d_qual, s_qual)
    del seq, qual
    log("[bench] rank %d: %d reads, %d k-mers, input staged in HBM in %.1fs; mode: %s" % (rank, n_reads, n_kmers, time.time() - t0, mode))

    stage = dict(hist1=0.0, scatter1=0.0, level2=0.0, bloom=0.0, commit=0.0, total=0.0)
    n_launch = 0
    xchg_s = 0.0

    # batch schedule: equal batches of batch_reads reads (batch boundaries never change results).  BFC_BENCH_RAMP=1 tries
    # smaller first batches for the cold filter.
    sched = []
    r = 0
    ramp = [8, 4, 2] if os.environ.get("BFC_BENCH_RAMP") else []  # measured: no gain (each batch streams the whole bitmap once), off by default
    for d in ramp:
        n = min(n_reads - r, (batch_reads // d) // 16 * 16)
        if n > 0:
            sched.append((r, r + n)); r += n
    while r < n_reads:
        n = min(n_reads - r, batch_reads)
        sched.append((r, r + n)); r += n
    n_batches = len(sched)
    if dist:
        import torch
        nb = torch.tensor([n_batches], device="cuda"); dist.all_reduce(nb, op=dist.ReduceOp.MAX); n_batches = int(nb.item())

    def step(acc):
        nonlocal n_launch, xchg_s
        g.reset()
        for t in range(n_batches):
            r0, r1 = sched[t] if t < len(sched) else (n_reads, n_reads)
            if eng is not None:
                tx = time.perf_counter()
                bdist.count_batch(eng, d_seq + r0 * stride, d_qual + r0 * stride, (r1 - r0) * stride)
                if acc:
                    xchg_s += time.perf_counter() - tx
            elif r1 > r0:
                g.count_dev(d_seq + r0 * stride, d_qual + r0 * stride, (r1 - r0) * stride)
            if acc and r1 > r0 and os.environ.get("BFC_BENCH_VERBOSE"):
                log("[bench] batch %d (%d reads), last finalised batch: %s" % (t, r1 - r0, {k_: round(v_, 3) for k_, v_ in g.last_batch_ms().items()}))

    def fence():
        g.sync()
        if dist:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    g.stage_ms(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    dt = time.perf_counter() - t0
    st = g.stats()
    tot, n_launch = g.stage_ms()
    stage.update(tot)
    # one more, untimed step with one kernel at a time: the same kernels' durations without the other stream's kernels beside them
    iso_bloom_ms = None
    if eng is None:
        os.environ["BFCG_SYNC_BATCHES"] = "1"
        g.stage_ms(reset=True)
        step(False)
        g.sync()
        iso, n_iso = g.stage_ms()
        iso_bloom_ms = iso["bloom"] / max(n_iso, 1)
        os.environ.pop("BFCG_SYNC_BATCHES", None)
    if dist:
        import torch
        t = torch.tensor([dt], device="cuda", dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
        tk = torch.tensor([float(n_kmers), float(st["n_kmers"]), float(st["n_seen"]), float(st["n_keys"])], device="cuda", dtype=torch.float64)
        dist.all_reduce(tk)
        total_kmers, gpu_kmers, tot_seen, tot_keys = [float(v) for v in tk.tolist()]
    else:
        total_kmers, gpu_kmers, tot_seen, tot_keys = float(n_kmers), float(st["n_kmers"]), float(st["n_seen"]), float(st["n_keys"])
    assert gpu_kmers == total_kmers, "GPU k-mer count %d != host count %d" % (gpu_kmers, total_kmers)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total_kmers * args.steps / dt / 1e6
        kmers_per_launch = n_kmers * args.steps / max(n_launch, 1)
        bloom_ms = stage["bloom"] / max(n_launch, 1)
        achieved = BLOOM_BYTES_PER_KMER * kmers_per_launch / (bloom_ms * 1e-3) / 1e9
        res = {
            "metric": "M k-mers/s counted (bloom+htab) on 150 bp reads", "value": round(value, 2), "unit": "M k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "c2: E. coli 100x 150 bp synthetic (bfcgen seed 2+rank, G=4.6M, 1%% subst.), k=31, -b33 -H4, l_pre 20, "
                                   "bloom-insert + htab build; 1 step = reset + full count of %d reads / %d k-mers per GPU" % (n_reads, n_kmers),
                       "batch_reads": batch_reads, "batches_per_step": n_batches,
                       "parallelism": mode, "exchange_plus_stages_s_per_step": round(xchg_s / args.steps, 4) if eng is not None else None,
                       "phase_cycles": st.get("phase_cycles"), "n_seen": int(tot_seen), "n_distinct": int(tot_keys), "slow_buckets": st["slow_buckets"], "tab_cshift": st["tab_cshift"],
                       "stage_ms_per_step": {kk: round(v / args.steps, 3) for kk, v in stage.items()}},
            "roofline": {"bound": "hbm", "kernel": "k_bloom (bloom regions in LDS + exact seen + table upsert)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": pmc_traffic(), "traffic_note": "HBM bytes per k_bloom launch from profiles/round1_pmc.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes of this command)", "kmers_per_launch": int(kmers_per_launch), "avg_launch_ms": round(bloom_ms, 4),
                         "algorithmic_bytes_per_kmer": BLOOM_BYTES_PER_KMER,
                         "isolated": None if not iso_bloom_ms else {"avg_launch_ms": round(iso_bloom_ms, 4), "achieved": round(BLOOM_BYTES_PER_KMER * kmers_per_launch / (iso_bloom_ms * 1e-3) / 1e9, 1),
                                                                    "frac": round(BLOOM_BYTES_PER_KMER * kmers_per_launch / (iso_bloom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                                    "note": "same kernel in one extra untimed step with one kernel at a time (in the timed region stage A / level 2 of the next batch run beside it on a second stream)"},
                         "whole_job_frac": round(BLOOM_BYTES_PER_KMER * total_kmers * args.steps / dt / 1e9 / HBM_PEAK_GBS / world, 4)},
        }
        if not args.no_cpu_baseline and world == 1:  # the reference on the host cores: at N=1 only
            try:
                cb = cpu_baseline(rs, os.cpu_count() or 1)
            except Exception as e:  # the baseline must never take the GPU number down with it
                log("[bench] cpu_baseline failed:", e)
                cb = None
            res["cpu_baseline"] = cb
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    g.dev_free(d_seq); g.dev_free(d_qual); g.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
