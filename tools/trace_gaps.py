#!/usr/bin/env python3
"""GPU idle-gap analysis of a rocprofv3 --kernel-trace csv: per bench step, kernel busy time vs wall span, and the
largest gaps with the kernels on either side.  Usage: trace_gaps.py <kernel_trace.csv> [n_last_steps]"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
# a step starts at the hash insert that follows x.cmap.drop_caches(): find the repeating k_round_minmax (one per encode)
enc_marks = [i for i, r in enumerate(rows) if r[2].startswith('k_round_minmax')]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
marks = enc_marks[-(steps + 1):]
tot_gap = collections.Counter(); n = 0
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a:b]
    busy = sum(e - s for s, e, _ in seg)
    span = seg[-1][1] - seg[0][0]
    gaps = []
    for (s0, e0, k0), (s1, e1, k1) in zip(seg[:-1], seg[1:]):
        g = s1 - e0
        if g > 0:
            gaps.append((g, k0[:40], k1[:40]))
    n += 1
    print(f'step: {len(seg)} kernels, busy {busy/1e6:.3f} ms, span {span/1e6:.3f} ms, idle {100*(1-busy/span):.1f}%')
    for g, k0, k1 in gaps:
        tot_gap[(k0, k1)] += g
print('largest idle gaps (mean us per step), between kernel A -> kernel B:')
for (k0, k1), g in tot_gap.most_common(25):
    print(f'{g/n/1e3:9.1f}  {k0}  ->  {k1}')
