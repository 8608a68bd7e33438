#!/bin/bash
# Build a variant of libvrt_hip.so with extra -D flags into build_variants/ (git-ignored; travels to the GPU box).
# usage: tools/build_variant.sh <name> [-DFLAG ...]      then: VRT_LIB_PATH=build_variants/libvrt_<name>.so python ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-value -Wno-unused-result "$@" \
  $ROOT/voxel_rt2_amd/csrc/vrt_kernels.hip $ROOT/voxel_rt2_amd/csrc/vrt_sky_kernels.hip $ROOT/voxel_rt2_amd/csrc/vrt_api.hip \
  -o $ROOT/build_variants/libvrt_$NAME.so
echo $ROOT/build_variants/libvrt_$NAME.so
