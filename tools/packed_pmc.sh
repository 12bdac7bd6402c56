#!/bin/bash
# SQ counter passes (rocprofv3 --pmc with --kernel-trace only) over tools/conv_packed_ab.py -> gpurun_out/packed_pmc/summary.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/packed_pmc; rm -rf $OUT /tmp/pp_*; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for LV in encoder decoder; do
rm -rf /tmp/pp_*
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/pp_$i -- python $R/tools/conv_packed_ab.py shell10 plain $LV > /tmp/pp_$i.log 2>&1
done
echo "== $LV level" >> $OUT/summary.txt
mkdir -p /tmp/pp_all_$LV; rm -rf /tmp/pp_all_$LV/*; mv /tmp/pp_[0-9]* /tmp/pp_all_$LV/ 2>/dev/null
python $R/tools/pmc_summary.py /tmp/pp_all_$LV k_conv_packed k_conv_gather_mfma >> $OUT/summary.txt 2>&1
tail -2 /tmp/pp_all_$LV/pp_1.log >> $OUT/summary.txt
done
cat $OUT/summary.txt
