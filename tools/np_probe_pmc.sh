#!/bin/bash
# Vector / scalar / LDS instructions per record of the product's no-payload decode kernel (k_inflate_par_np_lp) up to each cut-off of the inflate
# (the numbers of tools/par_probe_pmc.sh for the kernel the bench's decode leg runs).  Probe build, GPU box:
#   tools/variant.sh probe -DS5_PAR_PROBE; S5GPU_LIB=slow5tools_amd/_variants/libs5_probe.so tools/np_probe_pmc.sh [reads] [samples]
R=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-262144}; S=${2:-4000}
cd /tmp && export TMPDIR=/tmp
for cut in 11 12 13 14 1 5 6 2 7 3 9 0; do
  rm -rf /tmp/npp
  ( cd $R && S5_CUT=$cut rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d /tmp/npp -o pk -- python tools/decode_bulk.py $N $S np 2 ) > /tmp/npp.log 2>&1
  python3 - <<PY
import csv, glob, collections
names = {11: "block header, 3-bit lengths", 12: "+ code-length code tables", 13: "+ code-length sequence", 14: "+ lit/len symbols in canonical order", 1: "+ distance tables",
         5: "+ limits, lit/len lookup table", 6: "+ window, first (tail) pass", 2: "+ sync passes", 7: "+ output pass", 3: "+ runs, waiting matches", 9: "+ Adler-32 (inflate complete)", 0: "+ parse, svb-zd unpack (whole kernel)"}
f = glob.glob("/tmp/npp/**/*counter_collection.csv", recursive=True)[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "k_inflate_par_np" in r["Kernel_Name"]:
        acc.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
v = list(acc.values())[-1]
print("%-44s VALU/rec %8.0f SALU/rec %8.0f LDS/rec %7.0f" % (names[$cut], v.get("SQ_INSTS_VALU", 0) / $N, v.get("SQ_INSTS_SALU", 0) / $N, v.get("SQ_INSTS_LDS", 0) / $N))
PY
done
