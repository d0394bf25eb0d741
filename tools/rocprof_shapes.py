#!/usr/bin/env python3
"""Per (kernel, grid) average durations from a rocprofv3 rocpd .db (grid sizes in threads); with a second argument n also, per kernel, the average of its LAST n
launches (the timed steps of a bench.py run: the set-up launches the same kernels in other shapes -- ring fill, resets -- and comes first)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
q = """select s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x, count(*), avg(d.end-d.start)/1000.0, sum(d.end-d.start)/1000.0
from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id
group by s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z order by 8 desc"""
tot = 0
rows = list(c.execute(q))
tot = sum(r[7] for r in rows)
for r in rows:
    print(f"{r[0][:44]:44s} grid=({r[1]},{r[2]},{r[3]}) wg={r[4]:4d} n={r[5]:5d} avg={r[6]:8.1f}us  {100*r[7]/tot:5.1f}%")

if len(sys.argv) > 2:
    n = int(sys.argv[2])
    print(f"\nlast {n} launches of every kernel (the timed region):")
    per = {}
    for name, start, end in c.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start"):
        per.setdefault(name, []).append((end - start) / 1000.0)
    for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1][-n:])):
        if len(v) >= n:
            t = v[-n:]
            print(f"{name[:60]:60s} n={len(v):5d}  last {n}: avg={sum(t) / n:8.2f}us  min={min(t):8.2f}  max={max(t):8.2f}")
