#!/bin/bash
# PMC passes over tools/masked_probe.py (small target; --pmc with --kernel-trace only).  Output: gpurun_out/masked_pmc/*.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/masked_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU"; do
  i=$((i+1))
  for P in 0 1; do
    PCGC_PIPE=$P timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/mp_${i}_$P -- python $R/tools/masked_probe.py 5 > /tmp/mp_${i}_$P.log 2>&1
    python - <<PY > $OUT/set${i}_pipe$P.txt
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob('/tmp/mp_${i}_$P/*/*counter_collection.csv')[0])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r['Kernel_Name'][:60]
    if 'mfma' not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()): print(f'   {c:32s} {v:16.0f}')
PY
  done
done
tail -n +1 $OUT/*.txt
