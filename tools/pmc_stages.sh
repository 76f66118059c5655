#!/bin/bash
# instruction counts of k_encode_fused up to each stage cut-off (S5GPU_DEBUG_STAGE), one rocprofv3 pass each
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/pmcs
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for d in ${STAGES:-1 2 3 4 5 6 0}; do
  S5GPU_DEBUG_STAGE=$d timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH --output-format csv -d $OUT/s$d -o s$d -- python $R/tools/stage_time.py ${1:-200000} > $OUT/s$d.log 2>&1
done
python - <<PY
import csv, glob, collections
for d in [int(x) for x in "${STAGES:-1 2 3 4 5 6 0}".split()]:
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob("$OUT/s%d/*counter_collection.csv" % d):
        for row in csv.DictReader(open(f)):
            if "k_encode_fused" in row["Kernel_Name"]:
                acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    reads = ${1:-200000}
    print("stage<=%d  " % d + "  ".join("%s/read %.0f" % (k.replace("SQ_",""), acc[k]/n[k]/reads) for k in sorted(acc)))
PY
