#!/bin/bash
# zstd decode: the weights pass (k_zstd_weights) in front of the decoders, literal streams with a tail pass
O=gpurun_out/r04o; mkdir -p $O
( time timeout 900 python -m pytest tests/test_zstd.py -m gpu -x -q ) > $O/pytest_zstd.txt 2>&1; tail -n 6 $O/pytest_zstd.txt
tools/zstd_cuts.sh run 2>&1 | grep -v amdgpu | tee $O/zstd_cuts.txt
timeout 300 python tools/zstd_ref_frames.py > $O/zstd_ref_frames.txt 2>&1; tail -n 2 $O/zstd_ref_frames.txt
