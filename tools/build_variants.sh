#!/bin/bash
# Builds A/B variant libraries into tools/tmp/libse3d_<name>.so (never shipped; gitignored; they travel with gpurun): ONE translation unit is
# compiled with extra flags / from another source, the others are the tree's objects.
#   tools/build_variants.sh name "<extra hipcc flags>" [source file] [unit]     unit: stage1 (default) | api | stage2
set -e
cd "$(dirname "$0")/.."
CS=smartedgesensor3dhumanpose_amd/csrc
NAME=$1; EXTRA=$2; UNIT=${4:-stage1}; SRC=${3:-$CS/se3d_$UNIT.hip}
mkdir -p tools/tmp
make -C $CS > /dev/null
S2=""; [ $UNIT = stage2 ] && S2="-mllvm -disable-machine-licm -mllvm -sink-insts-to-avoid-spills"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -Wno-pass-failed -Iinclude -I$CS $S2 $EXTRA -c -o tools/tmp/${UNIT}_$NAME.o $SRC
OBJS=""
for u in api stage1 stage2 reproj stream; do
  if [ $u = $UNIT ]; then OBJS="$OBJS tools/tmp/${UNIT}_$NAME.o"; else OBJS="$OBJS $CS/se3d_$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o tools/tmp/libse3d_$NAME.so $OBJS
echo built tools/tmp/libse3d_$NAME.so
