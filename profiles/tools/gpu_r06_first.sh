#!/bin/bash
# round 6, first GPU call: does RCCL refuse (not hang on) two ranks on one GPU; the bench tests; the driver's exact bench command
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06a; mkdir -p $O
( RENI_SHARE_GPU=1 RENI_DIST_BACKEND=gloo timeout 180 python bench.py --gpus 2 --steps 2 --warmup 1 --batch 8 --no-cpu-baseline > $O/share2.out 2> $O/share2.err; echo "share2 rc=$?" ) 2>&1 | tee $O/share2.rc
tail -c 1500 $O/share2.out; tail -5 $O/share2.err
timeout 1500 python -m pytest tests/test_gpu_dist.py -x -q 2>&1 | tail -15 | tee $O/test_dist.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err; echo "bench rc=$?"
tail -n 1 $O/bench.out | wc -c; tail -n 1 $O/bench.out
