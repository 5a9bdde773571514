#!/bin/bash
# One rocprofv3 kernel-trace run of config 2 with the library RENI_HIP_LIB names; prints the tail kernels' average durations.
# Used as   gpu_variants.sh --rounds 1 --cmd "bash profiles/tools/gpu_dw1_ablate.sh" @base "-DRENI_EXP_DW1=1" ...
# (RENI_EXP_DW1, timing only: 1 = one g_1 load of eight, 2 = no dW GEMM, 4 = no sin).
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
D=gpurun_out/_dw1_$$
rocprofv3 --kernel-trace --stats -d $D -o k -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also > $D.log 2>&1
python3 profiles/summarize_rocpd.py $D/k_results.db $D.md > /dev/null 2>&1
grep "k_reni_dw1\|k_reduce_partials\|k_reni_train" $D.md | cut -c1-150
rm -rf $D $D.log $D.md
