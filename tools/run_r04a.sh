#!/bin/bash
# round 4, first GPU session: the suite (with the new NO_PAYLOAD soak tests), the tripwire variant, a long soak of the product build, the default bench line
O=gpurun_out/r04a; mkdir -p $O
{ nproc; free -g | head -2; df -h /dev/shm /tmp; rocm-smi --showmemuse 2>/dev/null | head -8; } > $O/box.txt 2>&1
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.txt 2>&1
T=slow5tools_amd/_variants/libs5_trip.so
S5GPU_LIB=$T timeout 600 python tools/np_tripwire.py 400 1000000 4000 default > $O/trip_default.txt 2>&1
S5GPU_LIB=$T timeout 600 python tools/np_tripwire.py 40 8192 4000 three > $O/trip_three.txt 2>&1
timeout 900 python tools/np_tripwire.py 3000 1000000 4000 default > $O/soak_product.txt 2>&1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/pytest.txt $O/trip_default.txt $O/trip_three.txt $O/soak_product.txt
