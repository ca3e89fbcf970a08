#!/bin/bash
# Experiment helper: rebuild ONE source with extra -D flags into tools/variants/libdcap_<tag>.so (the other objects come from
# the regular build).  Usage: tools/build_variant.sh <tag> <source.hip> <flags...>;  run with DCAP_LIB=<path>.
set -e
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift 2
mkdir -p tools/variants
obj=tools/variants/${src%.hip}_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -fPIC -std=c++17 -Wno-unused-result -Iinclude "$@" -c image-captioning_amd/csrc/$src -o $obj
others=$(ls image-captioning_amd/csrc/build/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/variants/libdcap_$tag.so $obj $others
echo tools/variants/libdcap_$tag.so
