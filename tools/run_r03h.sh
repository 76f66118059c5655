#!/bin/bash
# round 3 checkpoint: full GPU suite, default bench line, PMC traffic of every leg
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03h; mkdir -p $O
( time timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/pytest.log 2>&1; tail -6 $O/pytest.log
bash tools/pmc_traffic_all.sh $PWD/$O/pmc > $O/pmc_traffic.txt 2>&1; tail -8 $O/pmc_traffic.txt
cp $O/pmc/pmc_traffic.json profiles/pmc_traffic.json
( time python bench.py ) > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json; tail -3 $O/bench.err
