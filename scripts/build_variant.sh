#!/bin/bash
# A/B builds of the library with other compile-time constants (not shipped: build_ab/ is git-ignored but travels to the
# GPU box).  usage: scripts/build_variant.sh <name> <extra hipcc flags...>     e.g.  scripts/build_variant.sh la16 -DMODL_CD_LA=16
# Only the translation units of the four-wavefront solver are recompiled; the rest are the product's objects.
set -e
cd "$(dirname "$0")/../modl_amd/csrc"
NAME=$1; shift
OUT=../../build_ab/$NAME
mkdir -p $OUT
for f in cd_split cd_split_b cd_split_c cd_split_d; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c $f.hip -o $OUT/$f.o &
done
wait
OBJS=$(ls build/*.o | grep -v _diag | grep -v "build/cd_split")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmodl_hip.so $OBJS $OUT/cd_split.o $OUT/cd_split_b.o $OUT/cd_split_c.o $OUT/cd_split_d.o -ldl
ls -la $OUT/libmodl_hip.so
