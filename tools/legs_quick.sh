#!/bin/bash
# the mixed-lengths and long-read legs alone, for the product library and for variants: tools/legs_quick.sh [variant ...]
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
for v in default "$@"; do
  if [ $v = default ]; then unset S5GPU_LIB; else export S5GPU_LIB=$R/slow5tools_amd/_variants/libs5_$v.so; fi
  for mode in --mixed --long; do
    python bench.py $mode --steps 20 --cpu-seconds 0 --cpu-sweep-seconds 0 2>/dev/null | tail -n 1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-10s %-8s %8.1f GB/s  %8.3f ms/step  kernel_ms %s  B/sample %s' % ('$v', '$mode', d['value'], d['ms_per_step'], d.get('kernel_ms'), d.get('bytes_per_sample')))"
  done
done
