#!/usr/bin/env python3
"""Summarise a scripts/prof_round2.sh capture (rocprofv3 rocpd databases of `python bench.py`) for profiles/:

    python tools/make_round_md.py gpurun_out/prof_<tag> <workload> > profiles/round2_<workload>.md   (also writes profiles/round2_<workload>_pmc.json)

The json is what bench.py reads for `roofline.traffic`: HBM bytes per launch of every stage (FETCH_SIZE x 2 + WRITE_SIZE, the
gfx950 correction of MI355X_MICROARCH.md; counters in KiB) with the build id of the library the counters were measured on."""
import json
import os
import sys

ROUND = os.environ.get("ROUND", "4")

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_profile_md import kernel_rows, pmc_rows, short  # noqa: E402

STAGE = {"k_seg_setup": "scatter1", "k_hist1": "hist1", "k_colsum": "hist1", "k_scan_top": "hist1", "k_apply": "hist1", "k_scatter1": "scatter1", "k_scatter1_wc": "scatter1",
         "k_hist2": "level2", "k_scan2": "level2", "k_scatter2": "level2", "k_bloom": "bloom", "k_bloom3": "bloom",
         "k_commit": "commit", "k_commit_stream": "commit", "k_commit_seg": "commit"}


def bench_line(path):
    for line in open(path):
        if line.startswith('{"metric"'):
            return json.loads(line)
    return None


def table(rows):
    tot = sum(r[2] for r in rows)
    print("| kernel | calls | total ms | avg us | min us | max us | % of kernel time | wg | vgpr | sgpr | LDS B |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for name, n, t, mn, mx, wg, vg, sg, lds in rows:
        if t / tot < 0.0005:
            continue
        print("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.1f | %d | %s | %s | %s |" % (short(name), n, t / 1e6, t / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot, wg, vg, sg, lds))


def main(d, workload):
    j = bench_line(os.path.join(d, "trace.log"))
    print("# Round %s -- bench.py (workload %s) on one MI355X under rocprofv3\n" % (ROUND, workload))
    print("Command: `python bench.py --steps %d --warmup %d --no-cpu-baseline --no-verify --no-secondary` under `rocprofv3 --kernel-trace --stats` (scripts/prof_round2.sh)." % (j["steps"], j["warmup"]))
    print("Library build: `%s`.  Workload: %s\n" % (j.get("build_id"), j["config"]["workload"]))
    print("## As benchmarked: %.1f %s, %.3f ms/step; stage ms/step (HIP events inside the library) %s\n" % (j["value"], j["unit"], j["ms_per_step"], json.dumps(j["config"]["stage_ms_per_step"])))
    table(kernel_rows(os.path.join(d, "trace", "p_results.db")))
    ps = os.path.join(d, "trace_sync", "p_results.db")
    if os.path.exists(ps):
        j2 = bench_line(os.path.join(d, "trace_sync.log"))
        print("\n## Same command with BFCG_SYNC_BATCHES=1 (host waits for every batch: no overlap at all; %.1f %s, %.3f ms/step)\n" % (j2["value"], j2["unit"], j2["ms_per_step"]))
        table(kernel_rows(ps))
    pm = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        p = os.path.join(d, sub, "p_results.db")
        if os.path.exists(p):
            for k, v in pmc_rows(p).items():
                pm.setdefault(k, {}).update(v)
    if not pm:
        return
    jp = bench_line(os.path.join(d, "pmc_fetch.log")) or j
    # a batch = one launch of the bloom insert: k_bloom3, or k_bloom for a batch into an empty filter (copies resolved by class)
    n_batches = max(pm.get("k_bloom", {}).get("FETCH_SIZE", (0, 0))[0] + pm.get("k_bloom3", {}).get("FETCH_SIZE", (0, 0))[0], 1)
    print("\n## PMC counters (separate passes, BFCG_SYNC_BATCHES=1); FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; per launch of the kernel\n")
    print("| kernel | launches | read GB (FETCH_SIZE x2) | write GB | L2 hit % | SQ_WAIT_ANY / SQ_WAVE_CYCLES | LDS bank conflict cycles / LDS instruction |")
    print("|---|---|---|---|---|---|---|")
    stages = {}
    for k, c in sorted(pm.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", (0, 0))[1]):
        if "FETCH_SIZE" not in c:
            continue
        n = c["FETCH_SIZE"][0]
        f = 2 * c["FETCH_SIZE"][1] * 1024 / n
        w = c.get("WRITE_SIZE", (1, 0))[1] * 1024 / max(c.get("WRITE_SIZE", (1, 0))[0], 1)
        hit = c.get("TCC_HIT_sum", (1, 0))[1]; miss = c.get("TCC_MISS_sum", (1, 0))[1]
        wait = "%.2f" % (c["SQ_WAIT_ANY"][1] / max(c["SQ_WAVE_CYCLES"][1], 1)) if "SQ_WAIT_ANY" in c else ""
        bank = "%.2f" % (c["SQ_LDS_BANK_CONFLICT"][1] / max(c["SQ_INSTS_LDS"][1], 1)) if "SQ_LDS_BANK_CONFLICT" in c else ""
        if f + w > 1e6:
            print("| %s | %d | %.3f | %.3f | %.1f | %s | %s |" % (k, n, f / 1e9, w / 1e9, 100.0 * hit / max(hit + miss, 1), wait, bank))
        st = STAGE.get(k)
        if st:
            e = stages.setdefault(st, {"hbm_bytes_per_launch": 0.0, "read_bytes_per_launch": 0.0, "write_bytes_per_launch": 0.0, "kernels": []})
            # per BATCH of the library (= per launch of the stage's main kernel): helper kernels of a stage are folded in
            e["read_bytes_per_launch"] += f * n / n_batches; e["write_bytes_per_launch"] += w * n / n_batches
            e["hbm_bytes_per_launch"] += (f + w) * n / n_batches
            e["kernels"].append(k)
    out = {"workload": workload, "build_id": jp.get("build_id"), "batches": n_batches, "kmers_per_launch": jp["roofline"].get("kmers_per_launch"),
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum, separate passes of `python bench.py`, BFCG_SYNC_BATCHES=1; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024",
           "stages": stages}
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "round%s_%s_pmc.json" % (ROUND, workload)), "w") as f:
        json.dump(out, f, indent=1)
    print("\nPer batch and stage (what bench.py reports as `traffic`): " + ", ".join("%s %.2f GB" % (k, v["hbm_bytes_per_launch"] / 1e9) for k, v in stages.items()))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "c3")
