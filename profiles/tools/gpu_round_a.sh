#!/bin/bash
# usage: gpu_round_a.sh <tag> -- GPU suite, default bench line, one step's timeline at config 2 and at the curriculum's first stage
TAG=$1
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
python -m pytest tests -m gpu -q -x --deselect tests/test_api_cpu.py 2>&1 | tail -25 > $O/gpu_tests.log
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/kt -o k -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also > $O/kt.log 2>&1
python3 profiles/timeline_rocpd.py $O/kt/k_results.db > $O/timeline_c2.txt 2>&1
rm -rf $O/kt
rocprofv3 --kernel-trace --stats -d $O/kt -o k -- python3 bench.py --config c2_curric --steps 10 --warmup 2 --no-cpu-baseline > $O/kt2.log 2>&1
python3 profiles/timeline_rocpd.py $O/kt/k_results.db > $O/timeline_curric.txt 2>&1
rm -rf $O/kt
cat $O/gpu_tests.log; cat $O/timeline_c2.txt; cat $O/timeline_curric.txt; tail -3 $O/bench.err
