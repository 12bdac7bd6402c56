#!/usr/bin/env python3
"""Per-kernel summary of the TIMED steps of a bench run from a rocprofv3 --kernel-trace csv: every dispatch whose start lies inside
the window of the last N steps is counted, whatever its name (rocclr copy / fill kernels and torch's at::native kernels issued inside
a step included: filtering by timestamp, not by name).  A step boundary = the first k_round_minmax of an encode; `marks_per_step` =
how many of those one step issues (1 for frame, 8 for the batched blocks config, 7 for the sweep, ...).
Usage: trace_window_summary.py <kernel_trace.csv> [n_steps=5] [marks_per_step=1]"""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
per = int(sys.argv[3]) if len(sys.argv) > 3 else 1
marks = [i for i, r in enumerate(rows) if r[2].startswith('k_round_minmax')]
marks = marks[::per] if per > 1 else marks
if len(marks) < n_steps + 1:
    n_steps = len(marks) - 1
a, b = marks[-(n_steps + 1)], marks[-1]
seg = rows[a:b]
busy = sum(e - s for s, e, _ in seg)
span = seg[-1][1] - seg[0][0]
agg, cnt, mn, mx = collections.Counter(), collections.Counter(), {}, {}
for s, e, k in seg:
    d = e - s
    agg[k] += d; cnt[k] += 1
    mn[k] = min(mn.get(k, d), d); mx[k] = max(mx.get(k, d), d)
print(f'window: last {n_steps} steps ({per} encode mark(s) per step): {len(seg)} dispatches, {len(seg) / n_steps:.1f} per step; '
      f'GPU busy {busy / n_steps / 1e6:.3f} ms per step of {span / n_steps / 1e6:.3f} ms span (under the profiler)')
print(f'{"calls/step":>10} {"us/step":>10} {"avg_us":>9} {"min_us":>9} {"max_us":>9} {"pct":>6}  kernel')
for k, v in agg.most_common():
    print(f'{cnt[k] / n_steps:10.1f} {v / n_steps / 1e3:10.1f} {v / cnt[k] / 1e3:9.2f} {mn[k] / 1e3:9.2f} {mx[k] / 1e3:9.2f} {100 * v / busy:6.2f}  {k[:140]}')
