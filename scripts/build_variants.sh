#!/bin/bash
# Builds library variants for an A/B on the GPU box:  build_variants.sh name="-Dflags" ...  ->  build/libptamd_<name>.so
# (the default build is restored at the end)
set -e
cd "$(dirname "$0")/.."
mkdir -p build
for v in "$@"; do
  name=${v%%=*}; flags=${v#*=}; [ "$flags" = "$v" ] && flags=""
  rm -f cuda-pathtracer_amd/libptamd.so
  make -s lib EXTRA_HIPFLAGS="$flags"
  cp cuda-pathtracer_amd/libptamd.so build/libptamd_$name.so
  echo "built $name ($flags)"
done
rm -f cuda-pathtracer_amd/libptamd.so
make -s lib
