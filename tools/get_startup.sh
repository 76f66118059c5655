#!/bin/bash
# where a short `get` job's wall time goes: the process alone (exec + dynamic linking + static initialisers), the library's start-up, the job
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
[ -f /tmp/get_in.blow5 ] || python tools/get_bench.py 1000000 100000 > /dev/null 2>&1
TIMEFORMAT="%R s"
for i in 1 2 3; do echo -n "usage-only run (exec, ld.so, libamdhip64 static init): "; { time slow5tools_amd/s5get > /dev/null 2>&1; } 2>&1; done
for i in 1 2 3; do echo -n "get --benchmark 100k ids, K 4096: whole process "; { time slow5tools_amd/s5get --benchmark /tmp/get_in.blow5 /tmp/get_ids.txt 4096 8 > /tmp/get.out 2> /tmp/get.err; } 2>&1; grep 'reads of' /tmp/get.err | sed 's/^/    /'; grep 'GPU call' /tmp/get.err | sed 's/^/    /'; done
