#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db) as a per-kernel table (like --stats)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("""select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start)
  from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"""))
tot = sum(r[2] for r in rows) or 1
print(f"{'kernel':<86} {'calls':>6} {'total_ms':>11} {'avg_ms':>10} {'min_ms':>10} {'max_ms':>10} {'pct':>6}")
for r in rows:
    print(f"{r[0][:86]:<86} {r[1]:>6} {r[2]/1e6:>11.3f} {r[3]/1e6:>10.3f} {r[4]/1e6:>10.3f} {r[5]/1e6:>10.3f} {100*r[2]/tot:>6.2f}")
