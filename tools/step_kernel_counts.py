import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2].startswith('k_round_minmax')]
a, b = marks[-3], marks[-2]
seg = rows[a:b]
cnt = collections.Counter(); tim = collections.Counter()
for s, e, k in seg:
    cnt[k[:70]] += 1; tim[k[:70]] += e - s
print('kernels in one step:', len(seg), 'busy ms', sum(tim.values()) / 1e6)
for k, c in cnt.most_common(60):
    if c >= 3 or tim[k] > 20000: print(f'{c:4d} {tim[k]/1e3:8.1f} us  {k}')
