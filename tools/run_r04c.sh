#!/bin/bash
# round 4, third GPU session: the suite on the new kernels (two literals per step in the inflate, packed int16 pairs in the svb tile,
# encode-side launch order), A/B timings and instruction counts against the variants with each switched off
O=gpurun_out/r04c; mkdir -p $O
V=slow5tools_amd/_variants
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.txt 2>&1
for v in product two0; do
  L=; [ $v != product ] && L=$PWD/$V/libs5_$v.so
  S5GPU_LIB=$L python tools/decode_bulk.py 1000000 4000 np 8 > $O/bulk_np_$v.txt 2>&1
  S5GPU_LIB=$L python tools/decode_bulk.py 1000000 4000 full 6 > $O/bulk_full_$v.txt 2>&1
  S5GPU_LIB=$L python tools/par_decline_probe.py 2048 4000 262144 > $O/stock_$v.txt 2>&1
  S5GPU_LIB=$L python tools/decode_latency.py > $O/latency_$v.txt 2>&1
  S5GPU_LIB=$L KERNEL=k_inflate_par_np bash tools/pmc_kernel.sh python tools/decode_bulk.py 262144 4000 np 3 > $O/pmc_np_$v.txt 2>&1
done
for v in product pk0; do
  L=; [ $v != product ] && L=$PWD/$V/libs5_$v.so
  S5GPU_LIB=$L python tools/enc_stream_time.py > $O/enc_$v.txt 2>&1
  S5GPU_LIB=$L python tools/svb_stream_time.py > $O/svbs_$v.txt 2>&1
  S5GPU_LIB=$L KERNEL=k_encode_stream bash tools/pmc_kernel.sh python tools/enc_stream_time.py 400000 > $O/pmc_enc_$v.txt 2>&1
done
for f in $O/pytest.txt $O/bulk_np_*.txt $O/bulk_full_*.txt $O/stock_*.txt $O/enc_*.txt $O/pmc_np_*.txt $O/pmc_enc_*.txt; do echo "== $f"; tail -n 4 $f; done
