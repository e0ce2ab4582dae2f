#!/bin/bash
# A/B library: recompile ONE source with extra flags and link it with the other objects of the in-tree build.
#   scripts/exp/build_variant.sh <tag> <source.hip> <flags...>   ->  scripts/exp/lib_<tag>.so   (use with TC_LIB_PATH)
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
tag=$1; src=$2; shift 2
python -m transception_amd.build >/dev/null || exit 1
obj=$(mktemp --suffix=.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -munsafe-fp-atomics "$@" -c $ROOT/transception_amd/csrc/$src -o $obj || exit 1
others=$(ls $ROOT/transception_amd/build/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/scripts/exp/lib_$tag.so $obj $others && echo "built scripts/exp/lib_$tag.so"
rm -f $obj
