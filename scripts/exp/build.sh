#!/bin/bash
# experimental variants of the bridge attention forward: one .so per ATT_VAR value
cd "$(dirname "$0")"
for v in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=fast -munsafe-fp-atomics -DATT_VAR=$v -o libatt_v$v.so attn_exp.hip ../../transception_amd/csrc/attention.hip &
done
wait
