#!/usr/bin/env python3
"""Per-kernel PMC counter sums/averages from a rocprofv3 rocpd SQLite database (--pmc run)."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select s.kernel_name, p.name, count(*), sum(e.value) from rocpd_pmc_event e "
        "join rocpd_info_pmc p on e.pmc_id = p.id "
        "join rocpd_kernel_dispatch d on e.event_id = d.event_id "
        "join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name, p.name order by 1, 2").fetchall()
    last = None
    for k, c, n, v in rows:
        k = k.split("(")[0][:60]
        if k != last:
            print("==", k)
            last = k
        print("   %-28s calls=%-4d sum=%.4g avg=%.4g" % (c, n, v, v / n))


if __name__ == "__main__":
    main(sys.argv[1])
