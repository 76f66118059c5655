#!/bin/bash
# the whole GPU suite on the zstd decode changes
O=gpurun_out/r04p2; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/pytest.txt 2>&1; tail -n 6 $O/pytest.txt
