#!/bin/bash
# waiting matches by exact dependency range + both-ends copies + overlapped dword runs, as the product: the new chain test first, then the suite
O=gpurun_out/r04n; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "waiting_matches or awkward or stock_zlib" > $O/pytest_chain.txt 2>&1; tail -n 5 $O/pytest_chain.txt
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/pytest.txt 2>&1; tail -n 6 $O/pytest.txt
for rep in 1 2 3; do
python tools/par_decline_probe.py 2048 4000 262144 2>&1 | grep "inflate_par=1" >> $O/stock.txt
python tools/decode_bulk.py 1000000 4000 np 6 2>&1 | grep decode_bulk >> $O/stock.txt
done
cut -c1-200 $O/stock.txt
