#!/usr/bin/env python3
"""Summarise a scripts/prof_round.sh capture (rocprofv3 rocpd databases) as markdown + json for profiles/."""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def kernel_rows(path):
    db = sqlite3.connect(path)
    return db.execute(
        "select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start), max(d.workgroup_size_x), "
        "max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
        "on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()


def pmc_rows(path):
    db = sqlite3.connect(path)
    q = ("select s.kernel_name, p.name, count(distinct d.id), sum(e.value) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
         "join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by 1, 2")
    out = {}
    for k, c, n, v in db.execute(q):  # (instantiations of one kernel template share a row: launches and counters add up)
        e = out.setdefault(short(k), {})
        n0, v0 = e.get(c, (0, 0))
        e[c] = (n0 + n, v0 + v)
    return out


def short(name):
    import re
    m = re.match(r"_Z(?:N12_GLOBAL__N_1)?\d+(k_[a-z0-9_]+?)(I|P|N|v|E)", name)
    return m.group(1) if m else name.replace(".kd", "")


def main(d, title):
    rows = kernel_rows(os.path.join(d, "trace", "p_results.db"))
    tot = sum(r[2] for r in rows)
    print("# %s\n" % title)
    for line in open(os.path.join(d, "trace.log")):
        if line.startswith('{"metric"'):
            j = json.loads(line)
            print("Command: `python bench.py --steps %d --warmup %d --no-cpu-baseline` under `rocprofv3 --kernel-trace` (MI355X, gfx950).\n" % (j["steps"], j["warmup"]))
            print("bench line of that run: value = %.1f %s, %.3f ms/step, stage ms/step %s\n" % (j["value"], j["unit"], j["ms_per_step"], json.dumps(j["config"]["stage_ms_per_step"])))
    def table(rows, tot):
        print("| kernel | calls | total ms | avg us | min us | max us | % of kernel time | wg | vgpr | sgpr | LDS B |")
        print("|---|---|---|---|---|---|---|---|---|---|---|")
        for name, n, t, mn, mx, wg, vg, sg, lds in rows:
            print("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.1f | %d | %s | %s | %s |" % (short(name), n, t / 1e6, t / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot, wg, vg, sg, lds))
    print("## Kernel trace of the benchmarked command (batches pipelined over two streams: durations of overlapping kernels include the time they share the chip)\n")
    table(rows, tot)
    ps = os.path.join(d, "trace_sync", "p_results.db")
    if os.path.exists(ps):
        rows2 = kernel_rows(ps)
        for line in open(os.path.join(d, "trace_sync.log")):
            if line.startswith('{"metric"'):
                j = json.loads(line)
                print("\n## Same command with BFCG_SYNC_BATCHES=1 (no overlap between kernels; %.1f %s, %.3f ms/step)\n" % (j["value"], j["unit"], j["ms_per_step"]))
        table(rows2, sum(r[2] for r in rows2))
    pm = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        p = os.path.join(d, sub, "p_results.db")
        if os.path.exists(p):
            for k, v in pmc_rows(p).items():
                pm.setdefault(k, {}).update(v)
    print("\n## PMC counters per launch (separate passes, BFCG_SYNC_BATCHES=1 so that counters belong to one kernel)\n")
    print("FETCH_SIZE / WRITE_SIZE are KiB at the L2's memory side.  On gfx950 FETCH_SIZE counts a wide coalesced stream at half its")
    print("bytes (MI355X_MICROARCH.md, HBM section), so `read GB (x2)` doubles it; WRITE_SIZE is taken as is.\n")
    print("| kernel | launches | FETCH_SIZE KiB | WRITE_SIZE KiB | read GB (x2) | write GB | L2 hit % | SQ_WAIT_ANY / SQ_WAVE_CYCLES | LDS conflict / LDS instr |")
    print("|---|---|---|---|---|---|---|---|---|")
    summary = {}
    for k, c in sorted(pm.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", (0, 0))[1]):
        if "FETCH_SIZE" not in c:
            continue
        n = c["FETCH_SIZE"][0]
        f = c["FETCH_SIZE"][1] / n
        w = c.get("WRITE_SIZE", (1, 0))[1] / max(c.get("WRITE_SIZE", (1, 0))[0], 1)
        hit = c.get("TCC_HIT_sum", (1, 0))[1]
        miss = c.get("TCC_MISS_sum", (1, 0))[1]
        wa = c.get("SQ_WAIT_ANY", (1, 0))[1] / max(c.get("SQ_WAVE_CYCLES", (1, 1))[1], 1)
        lc = c.get("SQ_LDS_BANK_CONFLICT", (1, 0))[1] / max(c.get("SQ_INSTS_LDS", (1, 1))[1], 1)
        print("| %s | %d | %.4g | %.4g | %.3f | %.3f | %.1f | %.2f | %.2f |" % (k, n, f, w, 2 * f * 1024 / 1e9, w * 1024 / 1e9, 100.0 * hit / max(hit + miss, 1), wa, lc))
        summary[k] = dict(launches=n, fetch_kib=f, write_kib=w, hbm_bytes_per_launch=(2 * f + w) * 1024)
    json.dump(summary, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "round1_pmc.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "rocprofv3 summary")
