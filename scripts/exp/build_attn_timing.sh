#!/bin/bash
# timing builds of the hand-scheduled attention forward: one libattn_timing_<tag>.so per argument.
# An argument is "<tag>" or "<tag>:<ENV=VAL,ENV=VAL,...>" (generator knobs of gen_attn_asm.py); tag "base" = no ablation; any other bare tag is a TC_ATTN_ABLATE mode.
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
for arg in "$@"; do
  v=${arg%%:*}; envs=""
  [[ "$arg" == *:* ]] && envs=$(echo "${arg#*:}" | tr ',' ' ')
  d=$(mktemp -d)
  cp $ROOT/transception_amd/csrc/*.hip $ROOT/transception_amd/csrc/*.h $ROOT/transception_amd/csrc/*.inc $ROOT/transception_amd/csrc/gen_*.py $d/
  abl=""; [[ "$arg" != *:* && $v != base ]] && abl=$v
  defs=""; for e in $envs; do [[ $e == TC_AS_* ]] && defs="$defs -D$e"; done
  ( cd $d && env TC_ATTN_TIMING=${TC_ATTN_TIMING:-1} TC_ATTN_ABLATE=$abl $envs python gen_attn_asm.py && \
    sed -i 's|#include "../../include/transception_hip.h"|#include "'$ROOT'/include/transception_hip.h"|' tc_common.h && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=fast -munsafe-fp-atomics $defs -o $ROOT/scripts/exp/libattn_timing_$v.so attention_seg.hip attention.hip; rm -rf $d ) 2>&1 | grep -v hip-link &
done
wait
