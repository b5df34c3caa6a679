#!/usr/bin/env python
"""Summarise a rocprofv3 `--kernel-trace --stats` run (ROCm 7.2 writes a rocpd SQLite database)
into the per-kernel table committed under profiles/.

    python tools/rocpd_stats.py gpurun_out/prof_a/r1a_results.db profiles/r01a_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = list(c.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'calls', 'total_us', 'avg_us', 'percent'])
        for name, calls, tot, avg, pct in rows:
            w.writerow([name, calls, '%.3f' % tot, '%.3f' % avg, '%.3f' % pct])
    print('%d kernels -> %s' % (len(rows), out))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
