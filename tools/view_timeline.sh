#!/bin/bash
# the timeline (S5VIEW_TIMING) of one s5view run per direction on 1 M-read files in /dev/shm
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
python - <<PY
import os, sys, subprocess
sys.path.insert(0, "$R")
import numpy as np, torch
import bench_e2e as E
from slow5tools_amd import _lib, press
L = _lib.lib(); _lib.check(L.s5gpu_init(0), "init")
os.makedirs("/dev/shm/s5tl", exist_ok=True)
if not os.path.exists("/dev/shm/s5tl/in.blow5"):
    E.write_blow5("/dev/shm/s5tl/in.blow5", L, _lib, press, torch, "cuda:0", 1000000, 4000)
    E.view_run("/dev/shm/s5tl/in.blow5", "/dev/shm/s5tl/in.slow5", "none", "none", 3, {"S5VIEW_READERS": "4", "S5VIEW_CHUNK_MB": "32"}, 8.0)
PY
for dir in "in.slow5 out.blow5" "in.blow5 out2.blow5"; do
  set -- $dir
  echo "== s5view $1 -> $2"
  S5VIEW_TIMING=1 S5VIEW_READERS=${READERS:-8} slow5tools_amd/s5view /dev/shm/s5tl/$1 /dev/shm/s5tl/$2 zlib svb-zd 4096 3 2>&1 | grep '\[t\]\|pipeline'
done
rm -rf /dev/shm/s5tl/out*.blow5
