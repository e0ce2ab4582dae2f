#!/bin/bash
# A/B two builds of the library under rocprofv3 on the same box: scripts/ab_profile.sh libA.so libB.so  (paths relative to transception_amd/)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for L in "$@"; do
  TC_LIB_PATH=$R/transception_amd/$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$L -- python $R/bench.py --steps 10 --warmup 2 --no-cpu --no-attn-events > /dev/null 2>&1
  cp "$(find /tmp/p_$L -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/ab_$L.csv
done
