#!/bin/bash
# occupancy / LDS sensitivity of the parallel inflate: variants of kernels.hip timed on the bulk decode (1 M x 4000, NO_PAYLOAD) and K = 4096
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03b; mkdir -p $O
V=slow5tools_amd/_variants
for v in base wait256 pad2k pad4k pad6k waves5 waves4 waves8; do
  S5GPU_LIB=$V/libs5_$v.so python tools/decode_bulk.py 1000000 4000 np 4 2>&1 | tail -1 | sed "s/^/$v: /"
  S5GPU_LIB=$V/libs5_$v.so python tools/decode_bulk.py 4096 4000 np 8 2>&1 | tail -1 | sed "s/^/$v K4096: /"
done | tee $O/variants.txt
