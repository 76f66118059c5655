#!/bin/bash
# round 3: the waiting list of the parallel inflate sized by the signal press (256 entries for svb-zd records)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03i; mkdir -p $O
python tools/decode_bulk.py > $O/bulk.txt 2>&1; tail -12 $O/bulk.txt
S5GPU_LIB=$PWD/slow5tools_amd/_variants/libs5_probe.so python tools/par_decline_probe.py > $O/decline.txt 2>&1; tail -8 $O/decline.txt
S5GPU_LIB=$PWD/slow5tools_amd/_variants/libs5_probe.so python tools/par_fixture_probe.py 8192 > $O/fixture.txt 2>&1; tail -5 $O/fixture.txt
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -k "decode or inflate or parity or container or get or stock" ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
