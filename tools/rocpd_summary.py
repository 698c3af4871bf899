#!/usr/bin/env python
"""Kernel-time summary (the `--stats` table) from a rocprofv3 rocpd SQLite database.
usage: python tools/rocpd_summary.py results.db [> profiles/xxx_kernel_stats.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(accum_vgpr_count), "
                  "max(lds_size), max(scratch_size) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
print('%-110s %7s %12s %10s %10s %10s %5s %5s %7s %7s %6s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'min_us', 'max_us', 'vgpr', 'agpr', 'lds', 'scratch', '%'))
for r in rows:
    print('%-110s %7d %12.3f %10.1f %10.1f %10.1f %5d %5d %7d %7d %6.2f' % (r[0][:110], r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, r[6], r[7], r[8], r[9], 100.0 * r[2] / tot))
print('total kernel time: %.3f ms' % (tot / 1e6))
