#!/bin/bash
# Builds A/B variant libraries of se3d_stage1.hip into tools/tmp/libse3d_<name>.so (never shipped; gitignored; they travel with gpurun).
#   tools/build_variants.sh name "<extra hipcc flags>" [source file]     (source default: the tree's se3d_stage1.hip)
set -e
cd "$(dirname "$0")/.."
CS=smartedgesensor3dhumanpose_amd/csrc
NAME=$1; EXTRA=$2; SRC=${3:-$CS/se3d_stage1.hip}
mkdir -p tools/tmp
make -C $CS > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -Wno-pass-failed -Iinclude -I$CS $EXTRA -c -o tools/tmp/stage1_$NAME.o $SRC
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o tools/tmp/libse3d_$NAME.so tools/tmp/stage1_$NAME.o $CS/se3d_api.o $CS/se3d_stage2.o $CS/se3d_reproj.o $CS/se3d_stream.o
echo built tools/tmp/libse3d_$NAME.so
