#!/usr/bin/env python3
"""Dump the per-kernel summary (name, calls, total us, avg us, %) of a rocprofv3 --kernel-trace --stats
run (rocpd sqlite .db) as CSV.  Usage: tools/rocprof_summary.py <results.db> [out.csv]"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
    for name, calls, total, avg, pct in rows:
        w.writerow([name, calls, f"{total:.3f}", f"{avg:.3f}", f"{pct:.2f}"])


if __name__ == "__main__":
    main()
