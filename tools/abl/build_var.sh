#!/bin/bash
# usage: tools/abl/build_var.sh NAME "<flags for kvq_score_k.hip>" "<flags for kvq_mix_v.hip>" ["<flags for the rest>"]
# builds tools/abl/libkvq_NAME.so: a timing/experiment variant of the library (select it with KVQ_LIB=...)
cd "$(dirname "$0")/../.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-value -fvisibility=hidden -DKVQ_BUILD"
name=$1; d=/tmp/var_$name; mkdir -p $d
objs=""
for src in kvquant_amd/csrc/*.hip; do
  b=$(basename $src .hip); fl="$4"
  [ $b = kvq_score_k ] && fl="$2"
  [ $b = kvq_mix_v ] && fl="$3"
  [ $b = kvq_mix_v_api ] && fl="$3"
  [ $b = kvq_mix_v_wide ] && fl="$3"
  if [ -z "$fl" ]; then objs="$objs kvquant_amd/_obj/$b.o"; continue; fi
  /opt/rocm/bin/hipcc $F $fl -c $src -o $d/$b.o || exit 1
  objs="$objs $d/$b.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/abl/libkvq_$name.so $objs && echo built tools/abl/libkvq_$name.so
