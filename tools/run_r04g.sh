#!/bin/bash
# round 4, session 8: the encoder with one token pass (deflate_block MODE 3: per-lane bit strings) — the suite, A/B against the two-pass build, PMC
O=gpurun_out/r04g; mkdir -p $O
V=$PWD/slow5tools_amd/_variants
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.txt 2>&1
for v in product twopass; do
  L=; [ $v != product ] && L=$V/libs5_$v.so
  S5GPU_LIB=$L python tools/enc_stream_time.py > $O/enc_$v.txt 2>&1
  S5GPU_LIB=$L python tools/stream_time.py 1000000 > $O/stream_$v.txt 2>&1
  S5GPU_LIB=$L KERNEL=k_encode_stream bash tools/pmc_kernel.sh python tools/enc_stream_time.py 400000 > $O/pmc_enc_$v.txt 2>&1
done
python bench.py --no-legs --no-long --no-mixed --no-e2e --cpu-seconds 0 > $O/bench_quick.json 2> $O/bench_quick.err
tail -n 4 $O/pytest.txt; for f in $O/enc_*.txt $O/stream_*.txt $O/pmc_enc_*.txt; do echo "== $f"; grep -v amdgpu $f | tail -n 3 | cut -c1-600; done; cut -c1-300 $O/bench_quick.json
