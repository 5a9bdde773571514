#!/bin/bash
# the driver's command on the final sources (bench.py with the film_h256 sub-record), whole stdout kept
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06bench; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2.txt 2> $O/bench.err; echo "rc=$?"
tail -n 1 $O/bench_c2.txt > $O/bench_c2.json; wc -c $O/bench_c2.json; grep -c "^also " $O/bench_c2.txt
timeout 900 python -m pytest tests/test_gpu_dist.py -q -k bench 2>&1 | tail -3
