#!/bin/bash
# variant of the attention streams: scripts/exp/build_fwd_variant.sh <tag> <ahead> <defer 0|1>  ->  scripts/exp/lib_fwd_<tag>.so
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
tag=$1; ahead=$2; defer=$3
python -m transception_amd.build >/dev/null || exit 1
d=$(mktemp -d); cp $ROOT/transception_amd/csrc/*.hip $ROOT/transception_amd/csrc/*.h $ROOT/transception_amd/csrc/*.inc $ROOT/transception_amd/csrc/gen_*.py $d/
cd $d && TC_ATTN_AHEAD=$ahead TC_ATTN_DEFER=$defer python gen_attn_asm.py && TC_ATTN_AHEAD=$ahead python gen_dq_asm.py && \
sed -i 's|#include "../../include/transception_hip.h"|#include "'$ROOT'/include/transception_hip.h"|' tc_common.h && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -munsafe-fp-atomics -DTC_AS_AHEAD=$ahead -c attention_seg.hip -o as.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/scripts/exp/lib_fwd_$tag.so as.o $(ls $ROOT/transception_amd/build/*.o | grep -v /attention_seg.o) && echo "built lib_fwd_$tag"
rm -rf $d
