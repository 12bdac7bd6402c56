#!/usr/bin/env python3
"""Average PMC counters per kernel name from rocprofv3 counter_collection CSVs.  usage: pmc_summary.py dir [substr...]"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if len(sys.argv) > 2 and not any(s in k for s in sys.argv[2:]): continue
    print(k)
    for c, v in sorted(d.items()):
        print(f'    {c:32s} {sum(v) / len(v):16.1f}  (n={len(v)})')
