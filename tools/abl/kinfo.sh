#!/bin/bash
# usage: tools/abl/kinfo.sh "<extra flags>" [symbol regex] [source]  -- VGPRs / spills / scratch / LDS of the kernels of a
# source file as hipcc compiles it for gfx950 with the library's flags plus the extra ones
src=${3:-kvquant_amd/csrc/kvq_score_k.hip}
t=$(mktemp -d); r=$(pwd)
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-value -fvisibility=hidden -DKVQ_BUILD"
(cd $t && /opt/rocm/bin/hipcc $F $1 -I$r/kvquant_amd/csrc -I$r/include -save-temps -c $r/$src -o k.o 2>/dev/null)
awk -v pat="${2:-.}" '
/\.name:/ {name=$2}
/\.vgpr_count:/ {v=$2}
/\.vgpr_spill_count:/ {vs=$2}
/\.sgpr_spill_count:/ {ss=$2}
/\.private_segment_fixed_size:/ {p=$2}
/\.group_segment_fixed_size:/ {g=$2}
/\.wavefront_size:/ { if (name ~ pat) printf "%-64s vgpr %s spill %s sspill %s scratch %s lds %s\n", name, v, vs, ss, p, g }
' $t/*gfx950*.s
rm -rf $t
