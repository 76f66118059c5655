#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d /tmp/dp -o dp -- python $GRAFT_REPO_ROOT/tools/decode_latency.py > /tmp/dp.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/dp/**/*counter_collection.csv", recursive=True)[0]
acc=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "k_inflate" in r["Kernel_Name"]:
        key=(r["Dispatch_Id"], r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X"))
        acc.setdefault(key,{})[r["Counter_Name"]]=float(r["Counter_Value"])
for k,v in list(acc.items())[:13]:
    w=v.get("SQ_WAVES",1)
    print(k, "waves %d  cycles/wave %.0f  VALU/wave %.0f  SALU/wave %.0f  LDS/wave %.0f  BR/wave %.0f  wait/wave %.0f" % (w, 4*v["SQ_WAVE_CYCLES"]/w, v["SQ_INSTS_VALU"]/w, v["SQ_INSTS_SALU"]/w, v["SQ_INSTS_LDS"]/w, v["SQ_INSTS_BRANCH"]/w, 4*v["SQ_WAIT_ANY"]/w))
PY
