#!/bin/bash
# usage: tools/isa_snapshot.sh <out dir> [sources...]  -- gfx950 assembly of the kernel sources with the library's flags,
# debug / ident lines stripped, one .s per source: `diff -r` of two snapshots proves a source clean-up left the ISA alone
out=$1; shift
srcs=${@:-kvquant_amd/csrc/kvq_score_k.hip kvquant_amd/csrc/kvq_mix_v.hip kvquant_amd/csrc/kvq_fused_decode.hip kvquant_amd/csrc/kvq_fused_append.hip}
r=$(pwd); mkdir -p $out
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-value -fvisibility=hidden -DKVQ_BUILD"
for s in $srcs; do
  t=$(mktemp -d)
  (cd $t && /opt/rocm/bin/hipcc $F -I$r/kvquant_amd/csrc -I$r/include -save-temps -c $r/$s -o k.o 2>/dev/null)
  grep -v "^\s*;\|\.ident\|\.file\|\.loc\|^\s*\.p2align\|clang version" $t/*gfx950*.s | sed 's/;.*$//' > $out/$(basename $s .hip).s
  rm -rf $t
done
wc -l $out/*.s
