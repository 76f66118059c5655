#!/bin/bash
# k_compact with 16-byte copies (variant cmp16) against the product: mixed, long (one stream), two-pass headline; parity
O=gpurun_out/r04za; mkdir -p $O
V=$PWD/slow5tools_amd/_variants
for v in "$@"; do S5GPU_LIB=$V/libs5_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_ascii.py tests/test_container.py -m gpu -x -q 2>&1 | tail -2; done | tee $O/parity.txt
for rep in 1 2; do
for v in product "$@"; do
  L=$V/libs5_$v.so; [ $v = product ] && L=
  for m in "--mixed" "--long --long-streams 1" "--two-pass --no-long --no-mixed --no-e2e --no-legs"; do
    S5GPU_LIB=$L timeout 300 python bench.py $m --cpu-seconds 0 --cpu-sweep-seconds 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$v $m', d['value'], d['unit'], d.get('kernel_ms'))"
  done
done
done 2>&1 | tee $O/legs.txt
