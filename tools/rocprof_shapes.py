#!/usr/bin/env python3
"""Per (kernel, grid) average durations from a rocprofv3 rocpd .db (grid sizes in threads)."""
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
