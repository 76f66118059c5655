#!/bin/bash
# One GPU trip for a decoder change: bulk no-payload / full decode of 1 M own records, the K = 4096 get batch, instruction counts per cut-off (probe
# variant, if built), then the decode parity tests.  tools/dec_quick.sh OUTDIR [notest]
O=gpurun_out/${1:-dq}; mkdir -p $O
(python tools/decode_bulk.py 1000000 4000 np; python tools/decode_latency.py | tail -3; python tools/decode_bulk.py 1000000 4000 full 2>&1 | tail -1) 2>&1 | grep -v amdgpu.ids > $O/bulk.txt
cat $O/bulk.txt
if [ -f slow5tools_amd/_variants/libs5_probe.so ]; then
  S5GPU_LIB=slow5tools_amd/_variants/libs5_probe.so bash tools/par_probe_pmc.sh 262144 4000 2>&1 | awk '!seen[substr($0, 8)]++' > $O/pmc_cuts.txt; grep -E "cut-off|VALU" $O/pmc_cuts.txt
fi
[ "$2" = notest ] || timeout 1000 python -m pytest tests/test_gpu_parity.py tests/test_reference_fixtures.py tests/test_np_soak.py -m gpu -x -q 2>&1 | tail -3
