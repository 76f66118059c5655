#!/bin/bash
# rocprofv3 --kernel-trace --stats of one command; the kernel-stats CSV and the command's own stdout land in gpurun_out/<tag>/.
# usage: tools/kstats.sh <tag> <command...>     (run on the GPU box)
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout ${KSTATS_TIMEOUT:-600} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- "$@" ) > $OUT/stdout.txt 2> $OUT/stderr.txt
echo "rc=$?" >> $OUT/stderr.txt
f=$(ls $OUT/prof/*kernel_stats.csv $OUT/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $OUT/kernel_stats.csv
rm -rf $OUT/prof
tail -2 $OUT/stdout.txt | cut -c1-600
