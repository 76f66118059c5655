#!/bin/bash
# tools/hw_probe/pinned_small_d2h.sh [processes]: the stand-alone reproducer (no libslow5gpu) as many fresh processes, thread B's start swept over
# 0 .. 3 ms behind thread A's, small buffers of 4.9 KB (the library's case) and 64 KB, then the same with 2 MiB buffers (the library's cure).
cd "$(dirname "$0")/../.." || exit 1
N=${1:-300}
B=gpurun_out/pinned_small_d2h
[ -x "$B" ] || hipcc --offload-arch=gfx950 -O2 -o "$B" tools/hw_probe/pinned_small_d2h.hip -lpthread || exit 1
echo "# ROCm: $(cat /opt/rocm/.info/version 2>/dev/null)  kernel: $(uname -r)"
for pieces in 0 1; do
for size in 4900 65536 2097152; do
    bad=0
    for i in $(seq 1 "$N"); do
        "$B" $size $(( (i * 37) % 3000 )) 6 $pieces 2>> gpurun_out/pinned_small_d2h.err || bad=$((bad + 1))
    done
    echo "pinned buffer of $size bytes, $( [ $pieces = 1 ] && echo 'two small copies (520 + 16 bytes)' || echo 'one copy of the whole buffer' ): $bad bad of $N processes"
done
done
# shape 4: thread A's first launch comes out of a 1.2 MB code object and sets 64 function attributes (what the library's first batch does)
for size in 4900 2097152; do
    bad=0
    for i in $(seq 1 "$((2 * N))"); do
        "$B" $size $(( (i * 37) % 3000 )) 6 1 1 2>> gpurun_out/pinned_small_d2h.err || bad=$((bad + 1))
    done
    echo "pinned buffer of $size bytes, two small copies, big code object + attributes on thread A: $bad bad of $((2 * N)) processes"
done
echo "# stderr of the failing processes:"; sort gpurun_out/pinned_small_d2h.err 2>/dev/null | uniq -c | head -20
