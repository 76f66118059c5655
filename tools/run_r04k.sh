#!/bin/bash
# round 4: re-stamp the PMC traffic figures and the default bench line after the last host-side change of csrc (the kernels are untouched)
O=gpurun_out/r04k; mkdir -p $O
bash tools/pmc_traffic_all.sh $PWD/$O/pmc > $O/pmc_traffic.txt 2>&1
cp $O/pmc/pmc_traffic.json profiles/pmc_traffic.json
cp $O/pmc/pmc_traffic.json $O/pmc_traffic.json.txt
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.txt 2>&1
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -n 9 $O/pmc_traffic.txt; tail -n 4 $O/pytest.txt; tail -c 300 $O/bench_default.json
