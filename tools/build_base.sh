#!/bin/bash
# build_alt/libanihip_base.so from the csrc of a git revision (default HEAD): the baseline of an A/B against the working tree
set -e
cd "$(dirname "$0")/.."
REV=${1:-HEAD}
T=/tmp/anihip_base_src; rm -rf $T; mkdir -p $T/torchani_amd/csrc $T/include build_alt
for f in api nbr aev aev_generic mlp pair pack; do git show $REV:torchani_amd/csrc/$f.hip > $T/torchani_amd/csrc/$f.hip; done
git show $REV:torchani_amd/csrc/anihip_common.h > $T/torchani_amd/csrc/anihip_common.h
git show $REV:include/anihip.h > $T/include/anihip.h
hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -shared -o build_alt/libanihip_base.so $T/torchani_amd/csrc/*.hip 2>/dev/null
echo "built build_alt/libanihip_base.so from $REV"
