#!/bin/bash
# PMC passes over tools/irn_probe.py (small target; --pmc with --kernel-trace only).  Output: gpurun_out/irn_pmc/*.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/irn_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU" \
           "SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/ip_$i -- python $R/tools/irn_probe.py 5 ${1:-16} > /tmp/ip_$i.log 2>&1
  python - <<PY > $OUT/set$i.txt
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob('/tmp/ip_$i/*/*counter_collection.csv')[0])))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k = r['Kernel_Name'][:40]
    if 'k_irn' not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()): print(f'   {c:32s} {v:16.0f}')
PY
done
tail -n +1 $OUT/*.txt
