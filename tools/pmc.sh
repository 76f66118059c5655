#!/bin/bash
# PMC passes for the encode kernel (run on the GPU box): separate rocprofv3 runs per counter group.
# KERNEL=k_encode_stream (default, driven by tools/stream_time.py) or KERNEL=k_encode_fused (tools/stage_time.py)
R=$(cd "$(dirname "$0")/.." && pwd)
KERNEL=${KERNEL:-k_encode_stream}
if [ "$KERNEL" = k_encode_stream ]; then DRIVER=tools/stream_time.py; else DRIVER=tools/stage_time.py; fi
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p$i -- python $R/$DRIVER ${1:-200000} > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for row in csv.DictReader(open(f)):
        if "$KERNEL" in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    for k in acc: print("%-28s per-launch avg %16.1f  (%d launches)" % (k, acc[k] / n[k], n[k]))
PY
