#!/bin/bash
# build a variant of the library: build_variant.sh <out.so> <source.hip> <extra hipcc flags...>   (other objects from csrc/_obj)
set -e
cd "$(dirname "$0")/../.."
out=$1; src=$2; shift 2
base=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -ffp-contract=off "$@" -c rna_gan_amd/csrc/$src -o /tmp/variant_$base.o
objs=""
for o in rna_gan_amd/csrc/_obj/*.o; do
  if [ "$(basename $o .o)" == "$base" ]; then objs="$objs /tmp/variant_$base.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs
echo built $out
