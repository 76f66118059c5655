#!/bin/bash
# encoder variants against the product build, same session, interleaved: tools/run_r04s.sh NAME...   (variants built by tools/variant.sh)
O=gpurun_out/r04s; mkdir -p $O
V=$PWD/slow5tools_amd/_variants
for rep in 1 2 3; do
  python tools/enc_stream_time.py 2>&1 | tail -1
  for v in "$@"; do S5GPU_LIB=$V/libs5_$v.so python tools/enc_stream_time.py 2>&1 | tail -1; done
done | tee $O/enc_stream.txt
for v in "$@"; do
  echo "== $v: parity"; S5GPU_LIB=$V/libs5_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py -m gpu -x -q 2>&1 | tail -3
done | tee $O/parity.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in product "$@"; do
  L=$V/libs5_$v.so; [ $v = product ] && L=
  S5GPU_LIB=$L timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU --output-format csv -d $R/$O/pmc_$v -o p -- python $R/tools/stream_time.py 400000 > $R/$O/pmc_$v.log 2>&1
  python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$R/$O/pmc_$v/*counter_collection.csv") + glob.glob("$R/$O/pmc_$v/*/*counter_collection.csv")):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for row in csv.DictReader(open(f)):
        if "k_encode_stream" in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    print("$v:", "  ".join("%s %.0f" % (k[3:], acc[k] / n[k] / 400000) for k in sorted(acc)), "per read")
PY
done | tee $R/$O/pmc.txt
rm -rf $R/$O/pmc_*/
