#!/bin/bash
# end-of-round-6 profile set (run on the GPU box): PMC traffic of every leg's dominant kernel (stamped with the source hash), the default bench
# line, kernel stats of every bench mode (every `frac` of the line follows from a CSV here), the encoder's stage counters and PMC set, the bulk
# decode, the get start-up timeline, the staged legs; results under gpurun_out/r06p/.  Then, in the container: python tools/install_profiles.py r06p r06
cd "$(dirname "$0")/.." || exit 1
set -x
O=gpurun_out/r06p
rm -rf $O; mkdir -p $O
bash tools/pmc_traffic_all.sh $PWD/$O/pmc > $O/pmc_traffic.txt 2>&1
cp $O/pmc/pmc_traffic.json profiles/pmc_traffic.json          # (so that the bench lines below carry roofline.traffic)
cp $O/pmc/pmc_traffic.json $O/pmc_traffic.json.txt
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
python bench.py --decode > $O/bench_decode.json 2> $O/bench_decode.err
tools/kstats.sh r06p/enc python bench.py --no-long --no-mixed --no-e2e --cpu-seconds 0
tools/kstats.sh r06p/svb python bench.py --svb-only --cpu-seconds 0
tools/kstats.sh r06p/long python bench.py --long --cpu-seconds 0
tools/kstats.sh r06p/long_one_stream python bench.py --long --long-streams 1 --cpu-seconds 0
tools/kstats.sh r06p/mixed python bench.py --mixed --cpu-seconds 0
tools/kstats.sh r06p/decode python bench.py --decode --cpu-seconds 0
tools/kstats.sh r06p/decode_k4096_only python bench.py --decode --decode-batches-only --cpu-seconds 0
python tools/decode_bulk.py 1000000 4000 np 6 > $O/decode_bulk_np.txt 2>&1
python tools/decode_bulk.py 1000000 4000 full 6 > $O/decode_bulk_full.txt 2>&1
python tools/get_bench.py > $O/get_bench.txt 2>&1
bash tools/get_startup.sh > $O/get_startup.txt 2>&1
bash tools/legs_quick.sh > $O/legs_quick.txt 2>&1
STAGES="1 2 3 4 5 6 0" TAG=r06p_stages bash tools/stages.sh 400000 > $O/encode_stages.txt 2>&1
( echo "# tools/pmc.sh 400000 (KERNEL=k_encode_stream): per-launch averages over 400000 reads of 4000 samples; FETCH_SIZE / WRITE_SIZE in KiB"; KERNEL=k_encode_stream tools/pmc.sh 400000 ) > $O/pmc_k_encode_stream.txt 2>&1
( echo "# KERNEL=k_inflate_par_np tools/pmc_kernel.sh python tools/decode_bulk.py 262144 4000 np 3: totals over one launch of 262144 records"; KERNEL=k_inflate_par_np bash tools/pmc_kernel.sh python tools/decode_bulk.py 262144 4000 np 3 ) > $O/pmc_k_inflate_par_np.txt 2>&1
for v in probe; do [ -f slow5tools_amd/_variants/libs5_$v.so ] && S5GPU_LIB=slow5tools_amd/_variants/libs5_$v.so python tools/par_probe.py 262144 4000 > $O/par_probe_262144.txt 2>&1; done
python tools/par_decline_probe.py 2048 4000 262144 > $O/par_stock_zlib.txt 2>&1
# instruction counts per phase of the inflate (probe build): the slot form's kernel and the product's no-payload kernel; latency-mode cut-offs of a get batch
if [ -f slow5tools_amd/_variants/libs5_probe.so ]; then
  ( echo "# tools/par_probe_pmc.sh 262144 4000 (k_inflate_par<0>, probe build): vector / scalar / LDS instructions per record UP TO each cut-off";
    S5GPU_LIB=slow5tools_amd/_variants/libs5_probe.so bash tools/par_probe_pmc.sh 262144 4000 2>&1 | awk '!seen[substr($0, 8)]++';
    echo "# tools/np_probe_pmc.sh 262144 4000 (k_inflate_par_np_lp, the product's no-payload kernel)";
    S5GPU_LIB=slow5tools_amd/_variants/libs5_probe.so bash tools/np_probe_pmc.sh 262144 4000 2>&1;
    echo "# tools/par_probe.py 4096 4000: one get batch — the kernel's time is the slowest record's";
    S5GPU_LIB=slow5tools_amd/_variants/libs5_probe.so python tools/par_probe.py 4096 4000 2>&1 | grep -v amdgpu.ids | head -14 ) > $O/inflate_phase_counts.txt 2>&1
fi
( echo "# tools/np_lds_soak.py 24000 + tools/decode_soak.py 6000 on the final sources"; python tools/np_lds_soak.py 24000 2>&1 | grep -v amdgpu.ids; python tools/decode_soak.py 6000 2>&1 | grep -v amdgpu.ids ) > $O/decode_soak.txt 2>&1
rm -rf gpurun_out/r06p_stages gpurun_out/pmc
tail -9 $O/pmc_traffic.txt; tail -c 300 $O/bench_default.json
