#!/bin/bash
# HBM traffic of the bulk decode (k_inflate_par<true>, which also unpacks) under `bench.py --decode`: FETCH_SIZE and WRITE_SIZE in
# separate rocprofv3 passes (each fills the TCC counter budget), the launch with the largest grid = the 1 M-record call
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
echo "# tools/pmc_decode_traffic.sh: k_inflate_par<true> launch over 1000000 records of 4000 samples (bench.py --decode, bulk call); FETCH_SIZE / WRITE_SIZE in KiB"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pd_$c
  ( cd $R && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pd_$c -o pd -- python bench.py --decode --cpu-seconds 0 ) > /tmp/pd_$c.log 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("/tmp/pd_$c/**/*counter_collection.csv", recursive=True)[0]
best = None
for r in csv.DictReader(open(f)):
    if "k_inflate_par" in r["Kernel_Name"] and r["Counter_Name"] == "$c":
        g = int(r["Grid_Size"])
        if best is None or g > best[0]: best = (g, float(r["Counter_Value"]), r["Kernel_Name"][:40])
print("%-12s per-launch %16.1f  grid %d  %s" % ("$c", best[1], best[0], best[2]))
PY
done
