#!/bin/bash
# Memory-safety fuzzing of the host codec (csrc/hostcodec.cpp) under AddressSanitizer + UBSan, CPU only:
#   frame_decode_fuzz  — pcgc_frame_decode on randomly damaged _C/_F/_H/_num_points/_F.idx files (error or decode, never a bad access)
#   threads_tsan       — four threads through pcgc_frame_decode at once, under ThreadSanitizer
#   coders_fuzz        — range coder round trips over random tables / thread counts / checkpoint counts + corrupted streams; octree codec
# (round 3: this is how the "fractional range in a damaged _H.bin overruns the table buffer" bug was pinned down)
cd "$(dirname "$0")" || exit 1
set -e
for t in frame_decode_fuzz coders_fuzz; do
  g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -march=native $t.cpp ../../pcgcv2_amd/csrc/hostcodec.cpp -o /tmp/pcgc_$t -lz -lpthread
  /tmp/pcgc_$t ${1:-1000}
done
# four threads decoding four clouds at once, under ThreadSanitizer (shared pools, prewake, per-thread scratch)
g++ -O1 -g -fsanitize=thread -fno-omit-frame-pointer -std=c++17 -march=native threads_tsan.cpp ../../pcgcv2_amd/csrc/hostcodec.cpp -o /tmp/pcgc_threads_tsan -lz -lpthread
/tmp/pcgc_threads_tsan

