#!/bin/bash
# copy_record_out with 16-byte moves (variant) against the product: both stream kernels, k_pack (long / mixed legs); parity
O=gpurun_out/r04zb; mkdir -p $O
V=$PWD/slow5tools_amd/_variants
for v in "$@"; do S5GPU_LIB=$V/libs5_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_exzd.py -m gpu -x -q 2>&1 | tail -2; done | tee $O/parity.txt
for rep in 1 2 3; do
for v in product "$@"; do
  L=$V/libs5_$v.so; [ $v = product ] && L=
  S5GPU_LIB=$L python tools/enc_stream_time.py 2>&1 | tail -1
  S5GPU_LIB=$L python tools/svb_stream_time.py 2>&1 | grep -v amdgpu | tail -3
done
done 2>&1 | tee $O/stream.txt
for v in product "$@"; do
  L=$V/libs5_$v.so; [ $v = product ] && L=
  for m in "--mixed" "--long --long-streams 1"; do
    S5GPU_LIB=$L timeout 300 python bench.py $m --cpu-seconds 0 --cpu-sweep-seconds 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$v $m', d['value'], d['unit'], d.get('kernel_ms'))"
  done
done 2>&1 | tee $O/legs.txt
