#!/bin/bash
# LDS trims of the parallel inflate (distance table bits, run list, waiting list): own records, stock-zlib records, the reference's fixtures
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03c; mkdir -p $O
V=slow5tools_amd/_variants
for v in base d7 d7f64 d7f64w640 d7f64w576 d7f64w512; do
  echo "== $v"
  S5GPU_LIB=$V/libs5_$v.so python tools/decode_bulk.py 1000000 4000 np 4 2>&1 | tail -1
  S5GPU_LIB=$V/libs5_$v.so python tools/par_decline_probe.py 2048 4000 262144 2>&1 | grep "inflate_par="
  S5GPU_LIB=$V/libs5_$v.so python tools/par_fixture_probe.py 8192 2>&1 | grep -v amdgpu | head -12
done 2>&1 | tee $O/variants.txt
