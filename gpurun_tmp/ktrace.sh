#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dd
rocprofv3 --kernel-trace --output-format csv -d /tmp/dd -o dd -- python $GRAFT_REPO_ROOT/gpurun_tmp/dbg_decode.py > /tmp/dd.log 2>&1
tail -3 /tmp/dd.log
find /tmp/dd -name "*.csv" | head
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/dd/**/*kernel_trace.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if any(k in n for k in ("k_inflate","k_unpack","k_svbzd_decode")):
        print(n[:30], int(r["End_Timestamp"])-int(r["Start_Timestamp"]), "ns  grid", r.get("Grid_Size_X"), "wg", r.get("Workgroup_Size_X"))
PY
