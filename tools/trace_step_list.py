#!/usr/bin/env python3
"""One bench step of a rocprofv3 --kernel-trace csv as a list: start offset, duration, gap before, stream/queue, kernel.
Usage: trace_step_list.py <kernel_trace.csv> [step index from the end, default 2]"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', ''), r.get('Stream_Id', '')))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2].startswith('k_round_minmax')]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
a, b = marks[-k - 1], marks[-k]
# a step runs from the first kernel after the previous step's last decoder kernel: back up from the mark to the pyramid's fill
seg = rows[a:b]
t0 = seg[0][0]; prev_end = seg[0][0]
for s, e, name, q, st in seg:
    print(f'{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  q{q}/s{st}  {name[:90]}')
    prev_end = max(prev_end, e)
