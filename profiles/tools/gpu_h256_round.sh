#!/bin/bash
# usage: gpu_h256_round.sh <outdir>  -- kernel-trace summary of the non-headline variants (H = 256, FiLM, frozen)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
D=gpurun_out/_kt_$$
rocprofv3 --kernel-trace --stats -d $D -o k -- python3 profiles/tools/gpu_prof_variants.py > $O/prof.log 2>&1
python3 profiles/summarize_rocpd.py $D/k_results.db $O/kernel_stats_variants.md > /dev/null 2>&1 || ls -R $D | head
rm -rf $D
grep -v "at::native\|rocprim\|rocclr" $O/kernel_stats_variants.md | head -40
