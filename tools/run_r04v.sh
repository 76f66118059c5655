#!/bin/bash
# LZ77 long shape (k_deflate_lz<LzLong>) with 256 / 512 / 1024 threads per workgroup: time on 100 k-sample raw-signal records, sizes, parity
O=gpurun_out/r04v; mkdir -p $O
V=$PWD/slow5tools_amd/_variants
for v in "$@"; do
  echo "== $v"
  S5GPU_LIB=$V/libs5_$v.so timeout 600 python tools/lz_time.py 16384 100000 2>&1 | grep -v amdgpu.ids
  S5GPU_LIB=$V/libs5_$v.so timeout 600 python tools/lz_time.py 65536 4000 2>&1 | grep 'none + zlib'
  S5GPU_LIB=$V/libs5_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fixture or lz or LZ or none or solo or raw" 2>&1 | tail -2
done 2>&1 | tee $O/lz_tn.txt
