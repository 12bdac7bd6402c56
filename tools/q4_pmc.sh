#!/bin/bash
# PMC passes (rocprofv3 --pmc with --kernel-trace only, one counter set per run) over tools/child_q4_ab.py: the quad-block pass A next to
# the packed-N passes on the stride-1 candidates of a cloud.   usage: tools/q4_pmc.sh [cloud]   -> gpurun_out/q4_pmc/summary.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/q4_pmc; rm -rf $OUT /tmp/qp_*; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TCP|TA|TCC|TD|SQ|GRBM)_[A-Z0-9_]+(_sum|_avr)?\b" | sort -u > $OUT/counters_available.txt
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
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/qp_$i -- python $R/tools/child_q4_ab.py "$@" > /tmp/qp_$i.log 2>&1 || echo "set $i ($SET) failed: $(tail -2 /tmp/qp_$i.log)" >> $OUT/failed_sets.txt
done
python $R/tools/pmc_summary.py /tmp k_child > $OUT/summary.txt 2>&1
tail -4 /tmp/qp_1.log >> $OUT/summary.txt
cat $OUT/summary.txt
