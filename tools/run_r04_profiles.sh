#!/bin/bash
# end-of-round-4 profile set (run on the GPU box): PMC traffic of every leg's dominant kernel (stamped with the source hash), the default bench
# line (with the mixed leg, the e2e object and the PCIe-inclusive call), kernel stats of every bench mode — the long-read leg also on ONE stream
# and the decode leg also with the K = 4096 batches ALONE, so that every `frac` of the line follows from a CSV here —, instruction counts, the
# side tools' own outputs; results under gpurun_out/r04p/.  Then, back in the container: python tools/install_profiles.py r04p r04
cd "$(dirname "$0")/.." || exit 1
set -x
O=gpurun_out/r04p
rm -rf $O; mkdir -p $O
PROBE=$PWD/slow5tools_amd/_variants/libs5_probe.so     # tools/variant.sh probe -DS5_PAR_PROBE (cut-offs and counters of the parallel inflate)
bash tools/pmc_traffic_all.sh $PWD/$O/pmc > $O/pmc_traffic.txt 2>&1
cp $O/pmc/pmc_traffic.json profiles/pmc_traffic.json          # (so that the bench lines below carry roofline.traffic)
cp $O/pmc/pmc_traffic.json $O/pmc_traffic.json.txt
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
python bench.py --decode > $O/bench_decode.json 2> $O/bench_decode.err
tools/kstats.sh r04p/enc python bench.py --no-long --no-mixed --no-e2e --cpu-seconds 0
tools/kstats.sh r04p/svb python bench.py --svb-only --cpu-seconds 0
tools/kstats.sh r04p/long python bench.py --long --cpu-seconds 0
tools/kstats.sh r04p/long_one_stream python bench.py --long --long-streams 1 --cpu-seconds 0
tools/kstats.sh r04p/mixed python bench.py --mixed --cpu-seconds 0
tools/kstats.sh r04p/decode python bench.py --decode --cpu-seconds 0
tools/kstats.sh r04p/decode_k4096_only python bench.py --decode --decode-batches-only --cpu-seconds 0
tools/kstats.sh r04p/zstd python tools/zstd_time.py 1000000 4000
tools/kstats.sh r04p/lz python tools/lz_time.py 65536 4000
tools/kstats.sh r04p/mixed_decode python tools/mixed_lengths.py
python tools/svb_stream_time.py > $O/svb_stream_time.txt 2>&1
python tools/decode_bulk.py 1000000 4000 np 6 > $O/decode_bulk_np.txt 2>&1
python tools/decode_bulk.py 1000000 4000 full 6 > $O/decode_bulk_full.txt 2>&1
S5GPU_LIB=$PROBE python tools/par_probe.py 262144 4000 > $O/par_probe_262144.txt 2>&1
S5GPU_LIB=$PROBE python tools/par_probe.py 4096 4000 > $O/par_probe_4096.txt 2>&1
S5GPU_LIB=$PROBE python tools/par_decline_probe.py 2048 4000 262144 > $O/par_stock_zlib.txt 2>&1
S5GPU_LIB=$PROBE python tools/par_fixture_probe.py 8192 > $O/par_fixtures.txt 2>&1
python tools/exzd_time.py > $O/exzd_time.txt 2>&1
python tools/lz_time.py 16384 100000 > $O/lz_time_long_reads.txt 2>&1
python tools/e2e_probe.py 1000000 > $O/e2e_probe.txt 2>&1
python tools/get_bench.py > $O/get_bench.txt 2>&1
python tools/pcie_rate.py > $O/pcie_rate.txt 2>&1
bash tools/stages.sh > $O/encode_stages.txt 2>&1
bash tools/pmc_inflate.sh 65536 4000 > $O/pmc_k_inflate_par.txt 2>&1
( echo "# KERNEL=k_inflate_par_np tools/pmc_kernel.sh python tools/decode_bulk.py 262144 4000 np 3: totals over one launch of 262144 records (6144 persistent waves)"; KERNEL=k_inflate_par_np bash tools/pmc_kernel.sh python tools/decode_bulk.py 262144 4000 np 3 ) > $O/pmc_k_inflate_par_np.txt 2>&1
( echo "# tools/pmc.sh 400000 (KERNEL=k_encode_stream): per-launch averages over 400000 reads of 4000 samples; FETCH_SIZE / WRITE_SIZE in KiB"; KERNEL=k_encode_stream tools/pmc.sh 400000 ) > $O/pmc_k_encode_stream.txt 2>&1
tail -5 $O/pmc_traffic.txt; tail -c 300 $O/bench_default.json
