#!/bin/bash
# Build a variant of the library with extra -D flags for one source: scripts/build_variant.sh out.so gemm.hip -DGEMM_TR_A=0 ...
# (objects of the other sources come from transception_amd/build/; run python -m transception_amd.build first)
set -e
cd "$(dirname "$0")/.."
out=$1; src=$2; shift 2
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -munsafe-fp-atomics"
/opt/rocm/bin/hipcc $F "$@" -c transception_amd/csrc/$src -o /tmp/variant_${out%.so}_${src%.hip}.o
objs=""
for o in transception_amd/build/*.o; do
  if [ "$(basename $o)" = "${src%.hip}.o" ]; then objs="$objs /tmp/variant_${out%.so}_${src%.hip}.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o transception_amd/$out $objs
echo built transception_amd/$out
