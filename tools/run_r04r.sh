#!/bin/bash
# round 4, third session: two fused launches for mixed batches (option fused_tier2).  GPU suite, then bench.py --mixed over the
# second launch's budget (0 = one launch: the form of the earlier sessions) and the first launch's budget.
O=gpurun_out/r04r; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd - >/dev/null
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
for t2 in 0 16384 12288 10240 0 16384; do
  echo -n "fused_tier2=$t2 fused_cap=8192: "
  S5BENCH_OPTIONS=fused_tier2=$t2 timeout 300 python bench.py --mixed --cpu-seconds 0 --cpu-sweep-seconds 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['unit'], d.get('kernel_ms'), d.get('bytes_per_sample'))"
done 2>&1 | tee $O/mixed_tier2.txt
for c1 in 4096 6144 8192; do
  echo -n "fused_tier2=16384 fused_cap=$c1: "
  timeout 300 python bench.py --mixed --fused-cap $c1 --cpu-seconds 0 --cpu-sweep-seconds 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['unit'], d.get('kernel_ms'), d.get('bytes_per_sample'))"
done 2>&1 | tee -a $O/mixed_tier2.txt
tools/kstats.sh r04r/kstats_mixed python bench.py --mixed --cpu-seconds 0 --cpu-sweep-seconds 0
