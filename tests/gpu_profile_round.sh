#!/bin/bash
# usage: gpu_profile_round.sh <tag>  -- GPU test suite, bench line, rocprofv3 kernel stats and HBM-traffic PMC passes
TAG=$1
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
rocprofv3 --kernel-trace --stats -d $O/kt -o k -- python3 bench.py --steps 10 --warmup 2 > $O/kt.log 2>&1
python3 profiles/summarize_rocpd.py $O/kt/k_results.db $O/kernel_stats.md > /dev/null 2>&1 || ls -R $O/kt | head
rocprofv3 --kernel-trace --stats -d $O/ktv -o k -- python3 tests/gpu_prof_variants.py > $O/ktv.log 2>&1
python3 profiles/summarize_rocpd.py $O/ktv/k_results.db $O/variants_kernel_stats.md > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o p -- python3 bench.py --steps 3 --warmup 2 > $O/pmc_$c.log 2>&1
  python3 profiles/summarize_pmc.py $O/pmc_$c/p_results.db >> $O/pmc_counters.md 2>&1
done
python tests/gpu_perf_variants.py > $O/variants.txt 2>&1
rm -rf $O/kt $O/ktv $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
head -6 $O/kernel_stats.md; grep train_bf16 $O/pmc_counters.md; tail -12 $O/variants.txt
