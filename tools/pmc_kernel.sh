#!/bin/bash
# per-wave instruction counts of the kernels whose name contains $KERNEL, under any driver command (run on the GPU box):
#   KERNEL=k_unpack tools/pmc_kernel.sh python bench.py --decode --cpu-seconds 0
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk
( cd $R && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d /tmp/pk -o pk -- "$@" ) > /tmp/pk.log 2>&1
( cd $R && rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_INSTS_BRANCH --output-format csv -d /tmp/pk2 -o pk -- "$@" ) > /tmp/pk2.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("/tmp/pk", "/tmp/pk2"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "${KERNEL:-k_unpack}" in r["Kernel_Name"]:
            acc.setdefault((r["Kernel_Name"][:40], r["Dispatch_Id"], r["Grid_Size"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    best = None
    for k, v in acc.items():
        if best is None or int(k[2]) > int(best[0][2]): best = (k, v)
    if best:
        k, v = best
        print(k[0], "grid", k[2], " ".join("%s=%.4g" % kv for kv in v.items()))
PY
