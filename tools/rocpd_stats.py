#!/usr/bin/env python3
"""Per-kernel summary (count, total, average, min, max duration) from a rocprofv3 rocpd SQLite database
(`rocprofv3 --kernel-trace -d DIR -o NAME` writes NAME_results.db).  Used to produce profiles/*.md."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
        "max(d.grid_size_x*d.grid_size_y), max(d.workgroup_size_x), max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % | grid(threads) | wg | vgpr | sgpr | lds B |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, n, t, mn, mx, grid, wg, vg, sg, lds in rows:
        short = name.split("(")[0]
        if len(short) > 70:
            short = short[:70]
        print("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.1f | %s | %s | %s | %s | %s |" % (short, n, t / 1e6, t / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot, grid, wg, vg, sg, lds))


if __name__ == "__main__":
    main(sys.argv[1])
