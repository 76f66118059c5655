#!/bin/bash
# round 4, fourth GPU session: how fast a process writes into /dev/shm, the e2e knob sweep on the parallel writer, the default bench line,
# tripwire soak (lease 3)
O=gpurun_out/r04d; mkdir -p $O
gcc -O2 -pthread tools/hw_probe/shm_write_probe.c -o /tmp/shm_write_probe && /tmp/shm_write_probe /dev/shm/probe.bin 4 > $O/shm_write_probe.txt 2>&1
timeout 600 python tools/e2e_sweep.py 1000000 > $O/e2e_sweep.txt 2>&1
( time timeout 900 python bench.py --no-long > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
T=slow5tools_amd/_variants/libs5_trip.so
S5GPU_LIB=$T timeout 400 python tools/np_tripwire.py 4500 250000 4000 default > $O/trip_250k.txt 2>&1
S5GPU_LIB=$T timeout 300 python tools/np_tripwire.py 500 1000000 4000 default > $O/trip_1M.txt 2>&1
cat $O/shm_write_probe.txt; tail -n 25 $O/e2e_sweep.txt | cut -c1-330; tail -n 2 $O/trip_250k.txt; tail -n 2 $O/trip_1M.txt
