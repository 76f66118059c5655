#!/bin/bash
# round 3, first GPU session: full GPU test suite, the default bench line, bulk decode A/B (payload out / scratch), PMC traffic of both
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03a; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
( time timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/pytest.log 2>&1
tail -15 $O/pytest.log
( time python bench.py ) > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json; tail -3 $O/bench.err
python tools/decode_bulk.py 1000000 4000 np 6 > $O/bulk_np.txt 2>&1; tail -1 $O/bulk_np.txt
python tools/decode_bulk.py 1000000 4000 full 6 > $O/bulk_full.txt 2>&1; tail -1 $O/bulk_full.txt
MODE=np bash tools/pmc_decode_traffic.sh > $O/pmc_decode_np.txt 2>&1; cat $O/pmc_decode_np.txt
MODE=full bash tools/pmc_decode_traffic.sh > $O/pmc_decode_full.txt 2>&1; cat $O/pmc_decode_full.txt
