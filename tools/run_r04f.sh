#!/bin/bash
# round 4, sixth GPU session: the suite on the final sources, the inflate tunables once more (tail length of the first pass, length-part cadence,
# run-list size), s5view / s5get whole-process times with the fast exit, tripwire soak (lease 4)
O=gpurun_out/r04f; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.txt 2>&1
V=$PWD/slow5tools_amd/_variants
for v in product tail160 tail192 tail256 len2 len8 fill64; do
  L=; [ $v != product ] && L=$V/libs5_$v.so
  S5GPU_LIB=$L python tools/decode_bulk.py 1000000 4000 np 6 2>&1 | grep decode_bulk | sed "s/^/$v: /" >> $O/inflate_tunables.txt
  S5GPU_LIB=$L python tools/par_decline_probe.py 2048 4000 262144 2>&1 | grep "inflate_par=1" | sed "s/^/$v stock zlib: /" >> $O/inflate_tunables.txt
done
timeout 600 python tools/e2e_sweep.py 1000000 > $O/e2e_sweep.txt 2>&1
timeout 600 python tools/get_bench.py > $O/get_bench.txt 2>&1
T=$V/libs5_trip.so
S5GPU_LIB=$T timeout 300 python tools/np_tripwire.py 3000 250000 4000 default > $O/trip_250k.txt 2>&1
tail -n 3 $O/pytest.txt; cat $O/inflate_tunables.txt; grep -v amdgpu $O/e2e_sweep.txt | cut -c1-200 | tail -n 24; grep -v amdgpu $O/get_bench.txt | cut -c1-300; tail -n 1 $O/trip_250k.txt
