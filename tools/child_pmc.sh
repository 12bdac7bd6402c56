#!/bin/bash
# PMC passes (rocprofv3 --pmc with --kernel-trace only) over one configuration of tools/child_ab.py.
# usage: tools/child_pmc.sh <C 16|32> <waves> <ring>      -> gpurun_out/child_pmc/summary.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/child_pmc; rm -rf $OUT /tmp/cp_*; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/cp_$i -- python $R/tools/child_ab.py "$@" > /tmp/cp_$i.log 2>&1
done
python $R/tools/pmc_summary.py /tmp k_child k_conv_gather k_irn > $OUT/summary.txt 2>&1
tail -3 /tmp/cp_1.log >> $OUT/summary.txt
cat $OUT/summary.txt
