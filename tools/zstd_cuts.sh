#!/bin/bash
# where a zstd frame's decode time goes: variant builds that leave zstd_decode_wave early (S5_ZCUT = 1..5, csrc/zstd_dev.h; outputs are wrong by
# construction, only the time of the record stage is read).  Build here (tools/zstd_cuts.sh build), run on the GPU box (tools/zstd_cuts.sh run).
cd "$(dirname "$0")/.." || exit 1
if [ "$1" = build ]; then
  for c in 1 2 3 4 5; do tools/variant.sh zcut$c -DS5_ZCUT=$c || exit 1; done
  exit 0
fi
V=$PWD/slow5tools_amd/_variants
n=${2:-1000000}
for c in 1 2 3 4 5 full; do
  L=$V/libs5_zcut$c.so; [ $c = full ] && L=$PWD/slow5tools_amd/libslow5gpu.so
  echo -n "cut $c: "; S5GPU_LIB=$L python tools/zstd_time.py $n 4000 2>&1 | grep "zstd decode" | sed 's/ok=.*//'
done
echo "cuts: 1 frame, block and literals headers | 2 + tree description, ranks, Huffman table | 3 + literal streams | 4 + sequence tables | 5 + FSE chains of the sequences, no copies | full"
