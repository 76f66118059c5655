#!/bin/bash
# all-literal fast path as the product: GPU suite, headline kernel, mixed / long legs; phase clocks of the parallel inflate (variant iprobe)
O=gpurun_out/r04t; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.txt
for rep in 1 2 3; do python tools/enc_stream_time.py 2>&1 | tail -1; done | tee $O/enc_stream.txt
for m in --mixed --long; do
  timeout 300 python bench.py $m --cpu-seconds 0 --cpu-sweep-seconds 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$m', d['value'], d['unit'], d.get('kernel_ms'), d.get('bytes_per_sample'))"
done 2>&1 | tee $O/legs.txt
S5GPU_LIB=$PWD/slow5tools_amd/_variants/libs5_iprobe.so python tools/inflate_phases.py 262144 4000 2>&1 | tee $O/inflate_phases.txt
python tools/decode_bulk.py 1000000 4000 np 6 2>&1 | tail -1 | tee $O/decode_bulk_np.txt
