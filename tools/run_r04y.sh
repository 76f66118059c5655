#!/bin/bash
# second fused launch of a mixed batch with registers for four workgroups per CU (k_encode_fused<uint64_t, false, 4>)
O=gpurun_out/r04y; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mixed or overflow" 2>&1 | tail -2 | tee $O/pytest.txt
for t2 in 0 16384 12288 0 16384; do
  echo -n "fused_tier2=$t2: "
  S5BENCH_OPTIONS=fused_tier2=$t2 timeout 300 python bench.py --mixed --cpu-seconds 0 --cpu-sweep-seconds 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['unit'], d.get('kernel_ms'), d.get('bytes_per_sample'))"
done 2>&1 | tee $O/mixed_tier2.txt
S5BENCH_OPTIONS=fused_tier2=16384 tools/kstats.sh r04y/kstats python bench.py --mixed --cpu-seconds 0 --cpu-sweep-seconds 0
