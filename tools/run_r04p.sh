#!/bin/bash
# the whole GPU suite on the zstd decode changes, then the zstd rates and phase clocks
O=gpurun_out/r04p2; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/pytest.txt 2>&1; tail -n 6 $O/pytest.txt
for rep in 1 2 3; do timeout 300 python tools/zstd_time.py 1000000 4000 2>&1 | grep "zstd" >> $O/zstd_time.txt; timeout 300 python tools/zstd_ref_frames.py 2>&1 | grep k_zstd >> $O/zstd_time.txt; done; cat $O/zstd_time.txt
