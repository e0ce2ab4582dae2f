#!/bin/bash
# timing builds of the hand-scheduled dK/dV stream with parts removed (WRONG results): scripts/exp/lib_dkv_<tag>.so per argument
# tag = comma list for TC_ATTN_ABLATE / TC_DKV_ABLATE ("base" = nothing removed)
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
python -m transception_amd.build >/dev/null || exit 1
for v in "$@"; do
  ( d=$(mktemp -d); cp $ROOT/transception_amd/csrc/*.hip $ROOT/transception_amd/csrc/*.h $ROOT/transception_amd/csrc/*.inc $ROOT/transception_amd/csrc/gen_*.py $d/
    cd $d && abl=$([ $v = base ] && echo "" || echo $v) && TC_ATTN_ABLATE=$abl TC_DKV_ABLATE=$abl python gen_dkv_asm.py && \
    sed -i 's|#include "../../include/transception_hip.h"|#include "'$ROOT'/include/transception_hip.h"|' tc_common.h && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -munsafe-fp-atomics $(echo $v | grep -q b128stats && echo -DTC_DKV_B128STATS) -c attention_seg.hip -o as.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/scripts/exp/lib_dkv_$(echo $v | tr ',' '_').so as.o $(ls $ROOT/transception_amd/build/*.o | grep -v /attention_seg.o) && echo "built lib_dkv_$v"; rm -rf $d ) 2>&1 | grep -v hip-link &
done
wait
