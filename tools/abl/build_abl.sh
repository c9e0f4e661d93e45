#!/bin/bash
# builds ablation variants of libkvq.so (timing experiments only; results are wrong by construction)
cd "$(dirname "$0")/../.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-value -fvisibility=hidden -DKVQ_BUILD"
for n in "$@"; do
  /opt/rocm/bin/hipcc $F -DKVQ_ABL=$n -c kvquant_amd/csrc/kvq_score_k.hip -o /tmp/abl_score_$n.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/abl/libkvq_abl$n.so $(ls kvquant_amd/_obj/*.o | grep -v kvq_score_k.o) /tmp/abl_score_$n.o
done
