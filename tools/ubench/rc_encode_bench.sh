#!/bin/bash
# range encoder variants on synthetic latent-like symbols, compiled with the product's host compiler on the machine that runs them
cd "$(dirname "$0")/../.." || exit 1
CXX=/opt/rocm/lib/llvm/bin/clang++
for v in "" "-DPCGC_SINK_BYTES"; do
  $CXX -O3 -std=c++17 $v -Iinclude tools/ubench/rc_encode_bench.cpp pcgcv2_amd/csrc/hostcodec.cpp -o /tmp/rc_bench -lz -lpthread || exit 1
  echo "variant '$v':"; /tmp/rc_bench; /tmp/rc_bench
done
