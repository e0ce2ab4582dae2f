#!/bin/bash
# timing builds of the hand-scheduled attention forward: one libattn_timing_<ablation>.so per argument ("base" = no ablation)
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
for v in "$@"; do
  d=$(mktemp -d)
  cp $ROOT/transception_amd/csrc/*.hip $ROOT/transception_amd/csrc/*.h $ROOT/transception_amd/csrc/gen_attn_asm.py $d/
  ( cd $d && TC_ATTN_TIMING=1 TC_ATTN_ABLATE=$([ $v = base ] && echo "" || echo $v) python gen_attn_asm.py && \
    sed -i 's|#include "../../include/transception_hip.h"|#include "'$ROOT'/include/transception_hip.h"|' tc_common.h && \
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=fast -munsafe-fp-atomics -o $ROOT/scripts/exp/libattn_timing_$v.so attention_seg.hip attention.hip ) 2>&1 | grep -v hip-link &
done
wait
