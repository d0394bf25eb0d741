#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace run (rocpd .db): over the LAST n dispatches (by start time) prints the wall
time they span, the sum of their durations, the idle remainder, and the mean gap in front of every kernel name.
Usage: tools/rocprof_gaps.py <results.db> [n = 4000]"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
rows = list(c.execute("""select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
order by d.start"""))[-n:]
wall = (rows[-1][2] - rows[0][1]) / 1e3
busy = sum(e - s for _, s, e in rows) / 1e3
print(f"{len(rows)} dispatches: wall {wall:.1f} us, kernels {busy:.1f} us ({100 * busy / wall:.1f} %), idle {wall - busy:.1f} us = {(wall - busy) / len(rows):.2f} us per dispatch")
gap, dur, cnt = collections.Counter(), collections.Counter(), collections.Counter()
for (_, _, e0), (name, s1, e1) in zip(rows, rows[1:]):
    k = name.split("(")[0].replace("void ", "")[:48]
    gap[k] += max(0, s1 - e0) / 1e3; dur[k] += (e1 - s1) / 1e3; cnt[k] += 1
for k in sorted(cnt, key=lambda k: -dur[k]):
    print(f"{k:48s} n={cnt[k]:6d} avg {dur[k] / cnt[k]:7.2f} us   gap in front {gap[k] / cnt[k]:6.2f} us")
