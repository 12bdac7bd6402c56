#!/bin/bash
# PMC passes (rocprofv3 --pmc with --kernel-trace only, one counter set per run) over tools/rows32_ab.py: the encoder's C = 32 InceptionResNet
# block on the stride-2 level of a cloud.   usage: tools/rows32_pmc.sh [cloud] [rows|q4]   -> gpurun_out/rows32_pmc/summary_<impl>.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CLOUD=${1:-shell10}; IMPL=${2:-rows}
OUT=$R/gpurun_out/rows32_pmc; mkdir -p $OUT; rm -rf /tmp/rp_*
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/rp_$i -- python $R/tools/rows32_ab.py $CLOUD pmc $IMPL > /tmp/rp_$i.log 2>&1 || echo "set $i ($SET) failed: $(tail -2 /tmp/rp_$i.log)" >> $OUT/failed_sets_$IMPL.txt
done
python $R/tools/pmc_summary.py /tmp k_rows > $OUT/summary_$IMPL.txt 2>&1
tail -2 /tmp/rp_1.log >> $OUT/summary_$IMPL.txt
cat $OUT/summary_$IMPL.txt
