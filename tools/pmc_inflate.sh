#!/bin/bash
# instruction counts per record of the inflate kernels (run on the GPU box): tools/par_probe.py under rocprofv3 --pmc
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --output-format csv -d /tmp/dp -o dp -- python $R/tools/par_probe.py ${1:-65536} ${2:-4000} > /tmp/dp.log 2>&1
tail -4 /tmp/dp.log
python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/dp/**/*counter_collection.csv", recursive=True)[0]
acc=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "k_inflate" in r["Kernel_Name"]:
        acc.setdefault((r["Kernel_Name"][:40], r["Dispatch_Id"]),{})[r["Counter_Name"]]=float(r["Counter_Value"])
seen=set()
for (name,_),v in acc.items():
    w=v.get("SQ_WAVES",1)
    if (name,w) in seen or w < 64: continue
    seen.add((name,w))
    print("%-42s waves %7d  per wave: cycles %.0f  VALU %.0f  SALU %.0f  LDS %.0f  branch %.0f  lanes/VALU %.1f" % (name, w, 4*v["SQ_WAVE_CYCLES"]/w, v["SQ_INSTS_VALU"]/w, v["SQ_INSTS_SALU"]/w, v["SQ_INSTS_LDS"]/w, v["SQ_INSTS_BRANCH"]/w, v["SQ_THREAD_CYCLES_VALU"]/max(v["SQ_INSTS_VALU"],1)))   # = active lanes per VALU instruction (tools/pmc_lanes.sh: a full wave reads 64.0)
PY
