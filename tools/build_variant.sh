#!/bin/bash
# development aid: a copy of the library with one source rebuilt under extra -D flags
#   tools/build_variant.sh <name> <source.hip> "<flags>"   ->  dss_vae_amd/lib/dev_<name>.so   (use with DSSVAE_LIB=...)
set -e
cd "$(dirname "$0")/../dss_vae_amd/csrc"
name=$1; src=$2; flags=$3
mkdir -p build_$name
hipcc $flags --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -c $src -o build_$name/${src%.hip}.o
objs=""
for f in gemm pointwise gru skinny decode; do
  if [ "$f.hip" == "$src" ]; then objs="$objs build_$name/$f.o"; else objs="$objs build/$f.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/dev_$name.so $objs
echo built dss_vae_amd/lib/dev_$name.so
