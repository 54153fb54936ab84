#!/bin/bash
# A/B tuning: builds tools/alt/libsimamba_<name>.so with extra flags on ONE source file, the other objects as built.
#   tools/build_alt.sh <name> <source.hip> "<extra hipcc flags>"
set -e
cd "$(dirname "$0")/../si_mamba_amd/csrc"
name=$1; src=$2; flags=$3
mkdir -p ../../tools/alt
obj=/tmp/alt_${name}_$(basename ${src%.hip}).o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags -c $src -o $obj
objs=$(ls *.o | grep -v "^$(basename ${src%.hip}).o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/alt/libsimamba_${name}.so $objs $obj
echo built tools/alt/libsimamba_${name}.so
