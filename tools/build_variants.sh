#!/bin/bash
# Build libanihip variants for A/B runs inside ONE gpurun call (development only):
#   tools/build_variants.sh <tag> "<extra hipcc -D flags>" [<tag> "<flags>" ...]
# -> build_alt/libanihip_<tag>.so ; select at run time with TORCHANI_AMD_LIB=build_alt/libanihip_<tag>.so
set -e
cd "$(dirname "$0")/.."
SRC=torchani_amd/csrc
OBJ=/tmp/anihip_obj
mkdir -p $OBJ build_alt
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC"
for f in api nbr aev aev_generic mlp mlp_fused mlp_prep pair pack train; do
  if [ ! -f $OBJ/$f.o ] || [ $SRC/$f.hip -nt $OBJ/$f.o ] || [ $SRC/anihip_common.h -nt $OBJ/$f.o ] || [ $SRC/train.h -nt $OBJ/$f.o ] || [ $SRC/mlp_fused.h -nt $OBJ/$f.o ] || [ $SRC/mlp_prep.h -nt $OBJ/$f.o ] || [ include/anihip.h -nt $OBJ/$f.o ]; then
    hipcc $FLAGS -c $SRC/$f.hip -o $OBJ/$f.o 2>/dev/null &
  fi
done
wait
while [ $# -ge 2 ]; do
  tag=$1; defs=$2; shift 2
  objs=""
  for f in api nbr aev aev_generic mlp mlp_fused mlp_prep pair pack train; do
    if grep -q "ANIHIP_" <<< "$defs" && { [ "$f" = "${VARIANT_TU:-aev}" ] || [ "${VARIANT_TU:-aev}" = "all" ]; }; then
      hipcc $FLAGS $defs -c $SRC/$f.hip -o $OBJ/${f}_$tag.o 2>/dev/null
      objs="$objs $OBJ/${f}_$tag.o"
    else
      objs="$objs $OBJ/$f.o"
    fi
  done
  hipcc --offload-arch=gfx950 -shared -fPIC -o build_alt/libanihip_$tag.so $objs
  echo "built build_alt/libanihip_$tag.so ($defs)"
done
