#!/bin/bash
# staged DEFLATE kernel with 512 threads per workgroup (product) against 256 (variant st256): long-read leg, mixed leg, parity
O=gpurun_out/r04x; mkdir -p $O
V=$PWD/slow5tools_amd/_variants
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py -m gpu -x -q 2>&1 | tail -3 > $O/pytest.txt
for rep in 1; do
for v in product "$@"; do
  L=$V/libs5_$v.so; [ $v = product ] && L=
  for m in "--long --long-streams 1" "--long" "--mixed"; do
    S5GPU_LIB=$L timeout 300 python bench.py $m --cpu-seconds 0 --cpu-sweep-seconds 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$v $m', d['value'], d['unit'], d.get('kernel_ms'), d.get('bytes_per_sample'))"
  done
done
done 2>&1 | tee $O/legs.txt
