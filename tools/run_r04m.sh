#!/bin/bash
# waiting matches of the parallel inflate: exact dependency ranges (S5_IP_EXACT_DEP), both-ends copy (S5_IP_COPY2), overlapped dword runs
# (S5_IP_FILL2) against the product: own streams (bulk decode, 1 M x 4000) and stock-zlib records; three repeats each, interleaved
O=gpurun_out/r04m; mkdir -p $O
V=$PWD/slow5tools_amd/_variants
: > $O/variants2.txt
for rep in 1 2 3; do
  for v in product exactc all3; do
    L=$V/libs5_$v.so; [ $v = product ] && L=$PWD/slow5tools_amd/libslow5gpu.so
    S5GPU_LIB=$L timeout 300 python tools/decode_bulk.py 1000000 4000 np 6 2>&1 | grep decode_bulk | sed "s/^/$v: /" >> $O/variants2.txt
    S5GPU_LIB=$L timeout 300 python tools/par_decline_probe.py 2048 4000 262144 2>&1 | grep "inflate_par=1" | sed "s/^/$v stock zlib: /" >> $O/variants2.txt
  done
done
cut -c1-200 $O/variants2.txt
S5GPU_LIB=$V/libs5_all3.so timeout 900 python -m pytest tests -m gpu -x -q -k "inflate or zlib or stock or golden or fixture or decode" > $O/pytest_all3.txt 2>&1
tail -n 3 $O/pytest_all3.txt
