#!/bin/bash
# SQ counter passes over the C = 64 children-level kernels (tools/quant_probe.py as the workload) -> gpurun_out/c64_pmc.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
rm -rf /tmp/c64_*; cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/c64_$i -- python $R/tools/quant_probe.py > /tmp/c64_$i.log 2>&1
done
python $R/tools/pmc_summary.py /tmp k_child_irn > $R/gpurun_out/c64_pmc.txt 2>&1
