#!/bin/bash
# round 4, fifth GPU session: the suite on the current sources, where s5view's time goes (stage summary + library trace), the LDS-resident
# payload variant of the no-payload decode (time + HBM traffic), the mixed leg, the default line without the long legs
O=gpurun_out/r04e; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.txt 2>&1
timeout 600 python tools/e2e_probe.py 1000000 > $O/e2e_probe.txt 2>&1
V=$PWD/slow5tools_amd/_variants/libs5_ldspay.so
python tools/decode_bulk.py 1000000 4000 np 6 > $O/bulk_np_product.txt 2>&1
S5GPU_LIB=$V python tools/decode_bulk.py 1000000 4000 np 6 > $O/bulk_np_ldspay.txt 2>&1
MODE=np bash tools/pmc_decode_traffic.sh 262144 > $O/traffic_np_product.txt 2>&1
S5GPU_LIB=$V MODE=np bash tools/pmc_decode_traffic.sh 262144 > $O/traffic_np_ldspay.txt 2>&1
S5GPU_LIB=$V KERNEL=k_inflate_par_np bash tools/pmc_kernel.sh python tools/decode_bulk.py 262144 4000 np 3 > $O/pmc_np_ldspay.txt 2>&1
( time timeout 900 python bench.py --no-long --no-legs --cpu-seconds 0 --no-e2e > $O/bench_mixed_leg.json 2> $O/bench_mixed_leg.err ) 2> $O/bench_mixed_leg.time
tail -n 3 $O/pytest.txt; grep -v amdgpu $O/e2e_probe.txt | cut -c1-400 | tail -n 80; tail -n 2 $O/bulk_np_*.txt $O/traffic_np_*.txt
