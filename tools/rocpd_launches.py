#!/usr/bin/env python3
"""Durations of every launch of the kernels whose name contains <pattern>, in launch order, from a rocprofv3 rocpd database."""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
out = [round((e - b) / 1e6, 2) for n, b, e in rows if sys.argv[2] in n]
print(sys.argv[2], len(out), "launches, ms:", out)
