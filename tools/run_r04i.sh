#!/bin/bash
# round 4, last session: the suite and the default bench line on the final tree
O=gpurun_out/r04i; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -n 4 $O/pytest.txt; tail -n 1 $O/smoke.txt; tail -c 400 $O/bench_default.json; grep real $O/bench_default.err
