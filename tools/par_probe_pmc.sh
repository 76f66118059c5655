#!/bin/bash
# Vector / scalar / LDS instructions per record of the parallel inflate UP TO each of tools/par_probe.py's cut-offs (run on the GPU box, probe build):
#   tools/variant.sh probe -DS5_PAR_PROBE; S5GPU_LIB=slow5tools_amd/_variants/libs5_probe.so tools/par_probe_pmc.sh [reads] [samples]
# Every k_inflate_par dispatch of the run is listed in launch order (par_probe.py: 3 x whole kernel, counters, 3 x each cut-off 91 / 92 / 93 / whole).
R=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-262144}; S=${2:-4000}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ppp
( cd $R && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d /tmp/ppp -o pk -- python tools/par_probe.py $N $S ) > /tmp/ppp.log 2>&1
grep -E "inflate_par=|cut-off|sync passes" /tmp/ppp.log
python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/ppp/**/*counter_collection.csv", recursive=True)[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "k_inflate_par" in r["Kernel_Name"]:
        acc.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"][:44]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for (d, k), v in sorted(acc.items()):
    print("%6d %-44s VALU/rec %8.0f SALU/rec %8.0f LDS/rec %7.0f" % (d, k, v.get("SQ_INSTS_VALU", 0) / $N, v.get("SQ_INSTS_SALU", 0) / $N, v.get("SQ_INSTS_LDS", 0) / $N))
PY
