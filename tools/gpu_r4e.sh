#!/bin/bash
# round-4 call E: what each part of phase 5 costs (variants with a part removed; wrong numbers, timing only)
export TMPDIR=/tmp
for r in 1 2; do
for v in "" nostore noprev nogemm noall; do
  lib=""; [ -n "$v" ] && lib=$PWD/build_alt/libanihip_$v.so
  echo "== variant ${v:-product}"
  TORCHANI_AMD_LIB=$lib timeout 300 python tools/kbench.py --side 92 --reps 5 --stages mlp --mask on --compact 2>&1 | grep -v amdgpu.ids | tail -1
done
done
