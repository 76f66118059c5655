#!/bin/bash
# cumulative stage times + VALU/SALU/LDS instruction counts of k_encode_fused per stage cut-off (S5GPU_DEBUG_STAGE)
# cut-offs: 1 svb+payload | 21 byte loop | 22 neighbour breaks | 23 run classification | 2 reductions | 3 code lengths |
#           4 codes + code-length header + costs | 5 bit totals + scan | 6 token pack | 0 whole kernel
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/${TAG:-stages}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
N=${1:-400000}
for d in ${STAGES:-1 21 22 23 2 3 4 5 6 0}; do
  S5GPU_DEBUG_STAGE=$d python $R/tools/stage_time.py $N 2>/dev/null | tail -1
  S5GPU_DEBUG_STAGE=$d timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/s$d -o s$d -- python $R/tools/stage_time.py $N > $OUT/s$d.log 2>&1
done
python - <<PY
import csv, glob, collections
prev = None
for d in [int(x) for x in "${STAGES:-1 21 22 23 2 3 4 5 6 0}".split()]:
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob("$OUT/s%d/*counter_collection.csv" % d) + glob.glob("$OUT/s%d/*/*counter_collection.csv" % d):
        for row in csv.DictReader(open(f)):
            if "k_encode_fused" in row["Kernel_Name"]:
                acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    reads = $N
    cur = {k: acc[k] / n[k] / reads for k in acc}
    print("stage<=%-3d " % d + "  ".join("%s %.0f" % (k.replace("SQ_", ""), cur[k]) for k in sorted(cur)) + ("   | delta VALU %+.0f" % (cur.get("SQ_INSTS_VALU", 0) - prev.get("SQ_INSTS_VALU", 0)) if prev else ""))
    prev = cur
PY
