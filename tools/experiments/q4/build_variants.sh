#!/bin/bash
# experiment libraries of the quad-block kernel: tools/ubench/_bin/libq4_<name>.so = child_q4.hip compiled with the given -D switches
# usage: build_variants.sh name1:"-DFLAG ..." name2:"..."   (all in parallel)
R=$(cd "$(dirname "$0")/../../.." && pwd)
B=$R/tools/ubench/_bin
mkdir -p $B
g++ -O2 -fPIC -c $R/tools/experiments/q4/stub.cpp -o $B/q4_stub.o
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-result -Wno-unused-variable -Wno-unused-function $flags \
      -Rpass-analysis=kernel-resource-usage -x hip -c $R/pcgcv2_amd/csrc/child_q4.hip -o $B/q4_$name.o 2>&1 | grep -E "error|    VGPRs:|Occupancy|Spill: [1-9]" ;
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libq4_$name.so $B/q4_$name.o $B/q4_stub.o && echo "built $name" ) &
done
wait
