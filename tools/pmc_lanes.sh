#!/bin/bash
# settles how to read SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU (tools/hw_probe/valu_lanes_probe.hip): run on the GPU box
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 $R/tools/hw_probe/valu_lanes_probe.hip -o /tmp/valu_lanes_probe || exit 1
rm -rf /tmp/pl
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/pl -o pl -- /tmp/valu_lanes_probe > /tmp/pl.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pl/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print("%-40s SQ_INSTS_VALU/wave %.0f   SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU = %.2f   SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = %.2f" % (
        k, v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1), v["SQ_THREAD_CYCLES_VALU"] / max(v["SQ_INSTS_VALU"], 1), v["SQ_ACTIVE_INST_VALU"] / max(v["SQ_INSTS_VALU"], 1)))
PY
