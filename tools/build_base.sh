#!/bin/bash
# build_alt/libanihip_base.so from the csrc of a git revision (default HEAD): the baseline of an A/B against the working tree
set -e
cd "$(dirname "$0")/.."
REV=${1:-HEAD}
T=/tmp/anihip_base_src; rm -rf $T; mkdir -p $T/torchani_amd/csrc $T/include build_alt
for f in $(git ls-tree --name-only $REV torchani_amd/csrc/); do git show $REV:$f > $T/$f; done
git show $REV:include/anihip.h > $T/include/anihip.h
hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -shared -o build_alt/libanihip_base.so $T/torchani_amd/csrc/*.hip 2>/dev/null
echo "built build_alt/libanihip_base.so from $REV"
