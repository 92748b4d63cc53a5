#!/usr/bin/env python3
"""Per-kernel call count, average and minimum duration from a rocprofv3 results database (rocpd SQLite format).

    python tools/kernel_stats.py gpurun_out/prof_x/name_results.db
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
dispatch = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
symbol = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
q = (f"select s.kernel_name, s.arch_vgpr_count, s.group_segment_size, count(*), avg(d.end - d.start) / 1000.0, min(d.end - d.start) / 1000.0 "
     f"from {dispatch} d join {symbol} s on d.kernel_id = s.id group by s.kernel_name order by 5 desc")
print(f"{'kernel':72s} {'vgpr':>5s} {'lds':>6s} {'calls':>6s} {'avg us':>9s} {'min us':>9s}")
for name, vgpr, lds, n, avg, mn in cur.execute(q):
    print(f"{name[:72]:72s} {vgpr:5d} {lds:6d} {n:6d} {avg:9.2f} {mn:9.2f}")
