#!/bin/bash
# end-of-round profile set (run on the GPU box): kernel stats of every bench mode + PMC passes of k_encode_stream; results under gpurun_out/r02p/
set -x
mkdir -p gpurun_out/r02p
python bench.py > gpurun_out/r02p/bench_default.json 2> gpurun_out/r02p/bench_default.err
tools/kstats.sh r02p/enc python bench.py --no-long --cpu-seconds 0
tools/kstats.sh r02p/svb python bench.py --svb-only --cpu-seconds 0
tools/kstats.sh r02p/long python bench.py --long --cpu-seconds 0
tools/kstats.sh r02p/mixed python bench.py --mixed --cpu-seconds 0
tools/kstats.sh r02p/decode python bench.py --decode
tools/kstats.sh r02p/zstd python tools/zstd_time.py 1000000 4000
tools/kstats.sh r02p/lz python tools/lz_time.py 65536 4000
KERNEL=k_encode_stream tools/pmc.sh 400000 > gpurun_out/r02p/pmc_k_encode_stream.txt 2>&1
tail -30 gpurun_out/r02p/pmc_k_encode_stream.txt
