#!/bin/bash
# Tuning build of libsacv_hip.so with extra compiler defines, for same-session A/B runs (SACV_HIP_LIB=...):
#   bash scripts/build_variant.sh <name> "-DSACV_W64_DGRAD=0.65"   ->  gpurun_variants/libsacv_<name>.so
set -e
name=$1; defs=$2
root=$(cd $(dirname $0)/.. && pwd)
tmp=/tmp/sacv_variant_$name
rm -rf $tmp && mkdir -p $tmp
cp -r $root/simpleaicv-pytorch-imagenet-coco-training_amd/csrc $tmp/csrc
rm -rf $tmp/csrc/build
make -C $tmp/csrc -j8 CXXFLAGS="--offload-arch=gfx950 -O3 -fPIC -fvisibility=hidden -ffp-contract=off -std=c++17 -Wno-unused-result $defs" LIB=$tmp/libsacv_hip.so > $tmp/build.log 2>&1 || { tail -20 $tmp/build.log; exit 1; }
mkdir -p $root/gpurun_variants
cp $tmp/libsacv_hip.so $root/gpurun_variants/libsacv_$name.so
echo built gpurun_variants/libsacv_$name.so
