#!/usr/bin/env python3
"""List, in launch order, every kernel of ONE warmed-up bench step from a rocprofv3 --kernel-trace csv (start offset, duration, grid, name).
Usage: step_kernel_list.py <kernel_trace.csv> [substring-filter]"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Grid_Size', r.get('Grid_Size_X', '?'))))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2].startswith('k_round_minmax')]
a, b = marks[-3], marks[-2]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
t0 = rows[a][0]
for s, e, k, g in rows[a:b]:
    if flt in k:
        print(f'{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  grid {g:>9}  {k[:110]}')
