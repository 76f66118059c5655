#!/bin/bash
# Cumulative stage times of the zstd fused encode kernel (cut-offs in csrc/zstd_enc_dev.h, env S5GPU_DEBUG_STAGE):
# 9 payload only, 1 + histogram, 2 + code lengths, 31 + codes and stream bit counts (no tree description),
# 3 + tree description, 4 + headers, 0 everything (stream packing, copy-out).
# usage: tools/zstd_stages.sh [n_reads] [n_samples]
for st in 9 1 2 31 3 4 0; do
    echo -n "stage $st: "
    S5GPU_DEBUG_STAGE=$st python tools/zstd_time.py ${1:-262144} ${2:-4000} 2>&1 | grep "zstd encode"
done
