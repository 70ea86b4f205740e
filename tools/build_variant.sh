#!/bin/bash
# builds a diagnostic variant of libdgg_hip.so with extra -D flags applied to ONE translation unit:
#   tools/build_variant.sh <name> <file.hip> <flags...>   ->   tools/_bin/libdgg_<name>.so   (timing experiments only; never shipped)
set -e
cd "$(dirname "$0")/../learning-adaptive-neighborhoods-for-gnns_amd/csrc"
name=$1; src=$2; shift 2
mkdir -p ../../tools/_bin
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function "$@" -c $src -o /tmp/variant_$name.o
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o ../../tools/_bin/libdgg_$name.so $objs /tmp/variant_$name.o
echo built tools/_bin/libdgg_$name.so
