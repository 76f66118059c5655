#!/bin/bash
# second half of the round-3 profile set after the tools' kernel-name filter and max_pay_cap hint were corrected: the PMC traffic pass and
# the two side tools again, into the same directory (kernel sources unchanged: the hash in pmc_traffic.json must still match)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03p; mkdir -p $O
bash tools/pmc_traffic_all.sh $PWD/$O/pmc > $O/pmc_traffic.txt 2>&1
cp $O/pmc/pmc_traffic.json $O/pmc_traffic.json.txt
python tools/exzd_time.py > $O/exzd_time.txt 2>&1
tools/kstats.sh r03p/zstd python tools/zstd_time.py 1000000 4000
tail -8 $O/pmc_traffic.txt; grep -v amdgpu $O/exzd_time.txt
