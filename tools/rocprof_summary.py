#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats result into per-kernel statistics (calls, total, average, min, max, %).
Input: the rocpd sqlite .db, or the *_kernel_stats.csv written with --output-format csv.
Usage: rocprof_summary.py results.db|kernel_stats.csv [substring-to-skip ...]"""
import csv
import sqlite3
import sys


def load(path):
    if path.endswith('.csv'):
        rows = []
        for r in csv.DictReader(open(path)):
            rows.append((r['Name'], int(r['Calls']), float(r['TotalDurationNs']), float(r['AverageNs']), float(r['MinNs']), float(r['MaxNs'])))
        return sorted(rows, key=lambda r: -r[2])
    db = sqlite3.connect(path)
    return list(db.execute('select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) '
                           'from kernels group by name order by 3 desc'))


def main():
    skip = [a for a in sys.argv[2:] if not a.startswith('--')]
    rows = load(sys.argv[1])
    rows = [r for r in rows if not any(s in r[0] for s in skip)]
    total = sum(r[2] for r in rows) or 1
    print(f'{"calls":>7} {"total_ms":>10} {"avg_us":>10} {"min_us":>10} {"max_us":>10} {"pct":>6}  kernel')
    for name, n, tot, avg, mn, mx in rows:
        print(f'{n:7d} {tot / 1e6:10.3f} {avg / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} {100 * tot / total:6.2f}  {name[:150]}')
    print(f'TOTAL kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches')


if __name__ == '__main__':
    main()
