#!/bin/bash
# Build a variant of libvrt_hip.so with extra -D flags into build_variants/ (git-ignored; travels to the GPU box), with the
# shipped library's flags (voxel_rt2_amd/build.py).
# usage: tools/build_variant.sh <name> [-DFLAG ...]      then: VRT_LIB_PATH=build_variants/libvrt_<name>.so python ...
set -e
set -o pipefail
cd "$(dirname "$0")/.."
python -m voxel_rt2_amd.build --variant "$@" 2>&1 | { grep -v "not a recognized feature" || true; }
