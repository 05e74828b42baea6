#!/usr/bin/env python3
"""Summarise a scripts/prof_c3.sh capture (config c3 under rocprofv3) as markdown for profiles/."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_profile_md import kernel_rows, pmc_rows, short  # noqa: E402


def main(d):
    print("# Round 1 -- config c3 (human chr1-sized genome 248 Mbp x 30, 150 bp reads, k=33, -b35) on one MI355X under rocprofv3\n")
    print("Command: `python scripts/c3_run.py --b 35 --batch-reads 2097152 --digest 0` (49.6 M reads = 5.84 G k-mers staged in HBM, 24 batches of")
    print("2 097 152 reads = 236 M k-mers; 262 144 bloom regions of 16 KiB; the count table grows to 2^30 slots = 8 GiB for 300 M keys).\n")
    for name, title in (("trace", "as run (batches pipelined over two streams)"), ("trace_sync", "BFCG_SYNC_BATCHES=1 (one kernel at a time)")):
        for line in open(os.path.join(d, name + ".log")):
            if line.startswith('{"batch_reads"'):
                j = json.loads(line)
                print("## %s: %.3f s wall = %.2f G k-mers/s; stage ms %s; k_bloom at %.1f GB/s algorithmic (128 B per k-mer) = %.1f %% of 8 TB/s\n" %
                      (title, j["wall_s"], j["G_kmers_per_s"], json.dumps(j["stage_ms"]), j["bloom_GBps_algorithmic"], 100 * j["bloom_frac"]))
        rows = kernel_rows(os.path.join(d, name, "p_results.db"))
        tot = sum(r[2] for r in rows)
        print("| kernel | calls | total ms | avg us | min us | max us | % of kernel time | wg | vgpr | sgpr | LDS B |")
        print("|---|---|---|---|---|---|---|---|---|---|---|")
        for nm, n, t, mn, mx, wg, vg, sg, lds in rows:
            print("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.1f | %d | %s | %s | %s |" % (short(nm), n, t / 1e6, t / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot, wg, vg, sg, lds))
        print()
    pm = {}
    for sub in ("pmc_fetch", "pmc_write"):
        p = os.path.join(d, sub, "p_results.db")
        if os.path.exists(p):
            for k, v in pmc_rows(p).items():
                pm.setdefault(k, {}).update(v)
    print("## PMC counters per launch (separate passes, one kernel at a time); FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950\n")
    print("| kernel | launches | read GB (FETCH_SIZE x2) | write GB | L2 hit % | algorithmic GB (k_bloom: 128 B x 236 M k-mers) |")
    print("|---|---|---|---|---|---|")
    for k, c in sorted(pm.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", (0, 0))[1]):
        if "FETCH_SIZE" not in c or c["FETCH_SIZE"][0] < 8:
            continue
        n = c["FETCH_SIZE"][0]
        f = c["FETCH_SIZE"][1] / n
        w = c.get("WRITE_SIZE", (1, 0))[1] / max(c.get("WRITE_SIZE", (1, 0))[0], 1)
        hit = c.get("TCC_HIT_sum", (1, 0))[1]; miss = c.get("TCC_MISS_sum", (1, 0))[1]
        print("| %s | %d | %.3f | %.3f | %.1f | %s |" % (k, n, 2 * f * 1024 / 1e9, w * 1024 / 1e9, 100.0 * hit / max(hit + miss, 1), "30.2" if k == "k_bloom" else ""))


if __name__ == "__main__":
    main(sys.argv[1])
