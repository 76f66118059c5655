#!/bin/bash
# round 4, session 9: the record size published before the counting / packing passes (k_encode_stream) against the old place (-DS5_DEFL_LATE_SIZE)
O=gpurun_out/r04h; mkdir -p $O
V=$PWD/slow5tools_amd/_variants
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.txt 2>&1
for v in product latesize product latesize; do
  L=; [ $v != product ] && L=$V/libs5_$v.so
  S5GPU_LIB=$L python tools/enc_stream_time.py 2>&1 | grep k_encode_stream | sed "s/^/$v /" >> $O/enc.txt
done
python bench.py --no-legs --no-long --no-mixed --no-e2e --cpu-seconds 0 > $O/bench_quick.json 2> $O/bench_quick.err
tail -n 4 $O/pytest.txt; cat $O/enc.txt; cut -c1-300 $O/bench_quick.json
