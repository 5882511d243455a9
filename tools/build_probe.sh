#!/bin/bash
# Build a probe variant of the library: ONE source recompiled with extra flags, linked with the product objects.
# usage: tools/build_probe.sh <name> <source.hip> "<extra flags>"   ->  dis-yolo_amd/libdisyolo_<name>.so   (travels with gpurun;
# run with DISYOLO_LIB=dis-yolo_amd/libdisyolo_<name>.so)
set -e
name=$1; src=$2; flags=$3
cd "$(dirname "$0")/../dis-yolo_amd/csrc"
[ -n "$NO_MAKE" ] || make -j8 > /dev/null
obj=/tmp/probe_${name}_${src%.hip}.o
/opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags -c $src -o $obj
others=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdisyolo_${name}.so $obj $others
echo built dis-yolo_amd/libdisyolo_${name}.so
