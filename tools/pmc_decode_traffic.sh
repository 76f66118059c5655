#!/bin/bash
# HBM traffic of the bulk decode under tools/decode_bulk.py: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (each fills the TCC
# counter budget).  MODE=np (default): k_inflate_par_np, fields + signals out (S5GPU_DEC_NO_PAYLOAD); MODE=full: k_inflate_par<1 with the
# payload slots written out.  Averages over the launches of the run (all of them the 1 M-record call).
R=$(cd "$(dirname "$0")/.." && pwd)
MODE=${MODE:-np}
READS=${1:-1000000}
if [ "$MODE" = np ]; then KERN=k_inflate_par_np; else KERN="k_inflate_par<1"; fi
cd /tmp && export TMPDIR=/tmp
echo "# tools/pmc_decode_traffic.sh MODE=$MODE: $KERN over $READS records of 4000 samples (tools/decode_bulk.py); FETCH_SIZE / WRITE_SIZE in KiB per launch"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pd_$c
  ( cd $R && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pd_$c -o pd -- python tools/decode_bulk.py $READS 4000 $MODE 3 ) > /tmp/pd_$c.log 2>&1
  tail -1 /tmp/pd_$c.log | grep decode_bulk
  python3 - <<PY
import csv, glob
f = glob.glob("/tmp/pd_$c/**/*counter_collection.csv", recursive=True)[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "$KERN" in r["Kernel_Name"] and r["Counter_Name"] == "$c"]
print("%-12s per-launch avg %16.1f  (%d launches)  $KERN" % ("$c", sum(v) / max(len(v), 1), len(v)))
PY
done
