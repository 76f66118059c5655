#!/bin/bash
# build an experimental variant of the library: tools/variant.sh NAME -DFLAG...   -> slow5tools_amd/_variants/libs5_NAME.so
# (git-ignored; use with S5GPU_LIB=... python tools/stage_time.py).  Never part of the product build.
cd "$(dirname "$0")/.." || exit 1
name=$1; shift
V=slow5tools_amd/_variants
mkdir -p $V
C=slow5tools_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-sched-strategy=max-memory-clause "$@" -c $C/kernels.hip -o $V/kernels_$name.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libs5_$name.so $V/kernels_$name.o $C/host_api.o $C/ascii_kernels.o $C/ascii_api.o $C/slow5_compat.o $C/blow5_file.o && ls -la $V/libs5_$name.so
