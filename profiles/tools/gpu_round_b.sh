#!/bin/bash
# usage: gpu_round_b.sh <tag> -- GPU suite (all), default bench line, a config-2 step's timeline
TAG=$1
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/gpu_tests.log
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/kt -o k -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also > $O/kt.log 2>&1
python3 profiles/timeline_rocpd.py $O/kt/k_results.db > $O/timeline_c2.txt 2>&1
rm -rf $O/kt
cat $O/gpu_tests.log; cat $O/timeline_c2.txt; tail -3 $O/bench.err
