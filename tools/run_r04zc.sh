#!/bin/bash
# inflate window load with 16 bytes per lane (variant win16) against the product: K = 4096 batches, bulk decode, stock-zlib records; parity
O=gpurun_out/r04zc; mkdir -p $O
V=$PWD/slow5tools_amd/_variants
for v in "$@"; do S5GPU_LIB=$V/libs5_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py  -m gpu -x -q 2>&1 | tail -2; done | tee $O/parity.txt
for rep in 1 2 3; do
for v in product "$@"; do
  L=$V/libs5_$v.so; [ $v = product ] && L=
  echo -n "$v: "; S5GPU_LIB=$L python tools/decode_bulk.py 1000000 4000 np 6 2>&1 | tail -1
  echo -n "$v: "; S5GPU_LIB=$L timeout 300 python bench.py --decode --decode-batches-only --cpu-seconds 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print(d.get('batch_latency_ms'), d.get('kernel_ms_per_batch'), d.get('reads_per_s'))"
done
done 2>&1 | tee $O/decode.txt
