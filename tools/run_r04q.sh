#!/bin/bash
# zstd decode variants: chain inline / function, 4 or 3 waves per SIMD
O=gpurun_out/r04q; mkdir -p $O
V=$PWD/slow5tools_amd/_variants
for v in "$@"; do
  L=$V/libs5_$v.so
  echo -n "$v own: "; S5GPU_LIB=$L python tools/zstd_time.py 1000000 4000 2>&1 | grep "zstd decode" | sed 's/ok=.*//'
  echo -n "$v libzstd: "; S5GPU_LIB=$L python tools/zstd_ref_frames.py 2>&1 | grep "k_zstd_inflate"
done | tee $O/variants.txt
