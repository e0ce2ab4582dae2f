cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python scripts/bench_ffn.py
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_ffn -o p -- python scripts/bench_ffn.py --reps 3 > gpurun_out/pmc_ffn.log 2>&1
ls gpurun_out/pmc_ffn
