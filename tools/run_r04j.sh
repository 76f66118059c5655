#!/bin/bash
# round 4, session 11: code-generation flags of kernels.hip (scheduler strategy, post-RA scheduler, unroll threshold) on the two headline kernels
O=gpurun_out/r04j; mkdir -p $O
V=$PWD/slow5tools_amd/_variants
for v in product fl_default fl_maxilp fl_minreg fl_iterilp fl_nopostsched fl_unroll; do
  L=; [ $v != product ] && L=$V/libs5_$v.so
  S5GPU_LIB=$L python tools/enc_stream_time.py 2>&1 | grep k_encode_stream | sed "s/^/$v /" >> $O/flags.txt
  S5GPU_LIB=$L python tools/decode_bulk.py 1000000 4000 np 5 2>&1 | grep decode_bulk | sed "s/^/$v /" >> $O/flags.txt
done
cut -c1-200 $O/flags.txt
