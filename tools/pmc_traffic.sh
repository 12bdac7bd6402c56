#!/bin/bash
# HBM traffic of the gather kernels from rocprofv3 PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE --pmc passes (TCC slot limits), with --kernel-trace only.  Output: profiles/pmc_traffic.json
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_traffic
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-events --serving-frames 0 --no-extra > $OUT/$C.log 2>&1
done
python $R/tools/pmc_traffic.py $OUT $R/profiles/pmc_traffic.json
find $OUT -name '*.csv' -size +1M -delete      # gpurun_out/ is capped at 64 MiB
