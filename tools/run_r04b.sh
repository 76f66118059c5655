#!/bin/bash
# round 4, second GPU session: tripwire soak (lease 2), the default bench line with the e2e object, mixed-length encode with / without the launch order
O=gpurun_out/r04b; mkdir -p $O
T=slow5tools_amd/_variants/libs5_trip.so
S5GPU_LIB=$T timeout 400 python tools/np_tripwire.py 4000 250000 4000 default > $O/trip_250k.txt 2>&1
S5GPU_LIB=$T timeout 300 python tools/np_tripwire.py 400 1000000 4000 default > $O/trip_1M.txt 2>&1
S5GPU_LIB=$T timeout 200 python tools/np_tripwire.py 300 2048 4000 three > $O/trip_three.txt 2>&1
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
timeout 300 python bench.py --mixed > $O/bench_mixed.json 2> $O/bench_mixed.err
S5BENCH_OPTIONS=order_min=0 timeout 300 python bench.py --mixed > $O/bench_mixed_noorder.json 2> $O/bench_mixed_noorder.err
for f in $O/trip_250k.txt $O/trip_1M.txt $O/trip_three.txt $O/bench_default.time; do tail -n 3 $f; done
