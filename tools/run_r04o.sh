#!/bin/bash
# zstd decode: weights pass, literal streams with a tail pass, sequences executed 64 at a time
O=gpurun_out/r04o; mkdir -p $O
( time timeout 900 python -m pytest tests/test_zstd.py -m gpu -x -q ) > $O/pytest_zstd.txt 2>&1; tail -n 6 $O/pytest_zstd.txt
for rep in 1 2; do timeout 300 python tools/zstd_time.py 1000000 4000 2>&1 | grep "zstd decode" >> $O/zstd_time2.txt; done; cat $O/zstd_time2.txt
timeout 300 python tools/zstd_ref_frames.py > $O/zstd_ref_frames.txt 2>&1; tail -n 2 $O/zstd_ref_frames.txt
