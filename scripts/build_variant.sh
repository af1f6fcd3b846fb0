#!/bin/bash
# Tuning build of the library with extra compiler flags on ONE source: bash scripts/build_variant.sh NAME file.hip "-DX=1 ..."
# -> gpurun_variants/libsacv_NAME.so (use with SACV_HIP_LIB=...; the other objects come from the tree's build)
set -e
name=$1; src=$2; flags=$3
cd "$(dirname "$0")/../simpleaicv-pytorch-imagenet-coco-training_amd/csrc"
make -s
mkdir -p build_var ../../gpurun_variants
obj=build_var/${src%.hip}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -fvisibility=hidden -ffp-contract=off -std=c++17 -Wno-unused-result $flags -c $src -o $obj
others=$(ls build/*.o | grep -v "build/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map -o ../../gpurun_variants/libsacv_$name.so $obj $others -ldl
echo built gpurun_variants/libsacv_$name.so
