#!/bin/bash
# like variant.sh, but the caller names ALL code-generation flags behind -O3 (variant.sh always adds the product's scheduler flag):
#   tools/variant_flags.sh NAME [flags...]  -> slow5tools_amd/_variants/libs5_NAME.so
cd "$(dirname "$0")/.." || exit 1
name=$1; shift
V=slow5tools_amd/_variants
mkdir -p $V
C=slow5tools_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $C/kernels.hip -o $V/kernels_$name.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libs5_$name.so $V/kernels_$name.o $C/host_api.o $C/ascii_kernels.o $C/ascii_api.o $C/slow5_compat.o $C/blow5_file.o && ls -la $V/libs5_$name.so
rm -f $V/kernels_$name.o
