#!/bin/bash
# tools/build_alt.sh <name> "<extra hipcc flags>" <source> [<source> ...] -- a variant of the library for A/B runs on one box
# (tools/ab_libs.sh, tools/r06_*_ab.sh): the named sources of fem_amd/csrc are compiled with the extra flags (-D switches of an
# experiment), every other object comes from the regular build, the result is fem_amd/lib/alt_<name>.so (git-ignored).
set -eo pipefail
name=$1; flags=$2; shift 2
cd "$(dirname "$0")/../fem_amd/csrc"
make -s -j8
B=../../build/csrc; A=../../build/alt_$name; mkdir -p $A
objs=""
for o in $B/*.o; do
  base=$(basename $o .o); keep=1
  for s in "$@"; do [ "$s" == "$base" ] && keep=0; done
  [ $keep == 1 ] && objs="$objs $o"
done
for s in "$@"; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result $flags -c $s -o $A/$s.o &
done
wait
for s in "$@"; do objs="$objs $A/$s.o"; done
hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/alt_$name.so $objs -ldl -lpthread
echo "built fem_amd/lib/alt_$name.so ($flags: $*)"
