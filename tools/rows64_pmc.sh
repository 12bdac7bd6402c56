#!/bin/bash
# PMC passes (rocprofv3 --pmc with --kernel-trace only, one counter set per run) over tools/rows64_run.py: k_rows_irn_a64 / _b64 on the two levels
# they serve in a vox10 frame.   usage: tools/rows64_pmc.sh [cloud]   -> gpurun_out/rows64_pmc/summary.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CLOUD=${1:-shell10}
OUT=$R/gpurun_out/rows64_pmc; rm -rf $OUT; mkdir -p $OUT; rm -rf /tmp/r64_*
cd /tmp && export TMPDIR=/tmp
python $R/tools/rows64_run.py $CLOUD time > $OUT/timing.txt 2>&1
for LV in enc dec; do
rm -rf /tmp/r64_*
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/r64_$i -- python $R/tools/rows64_run.py $CLOUD pmc $LV > /tmp/r64_$i.log 2>&1 || echo "set $i ($SET) failed" >> $OUT/failed_sets.txt
done
echo "== level: $LV" >> $OUT/summary.txt
python $R/tools/pmc_summary.py /tmp k_rows_irn >> $OUT/summary.txt 2>&1
done
cat $OUT/timing.txt $OUT/summary.txt
