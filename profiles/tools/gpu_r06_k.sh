#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06k; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/tests.txt
python bench.py --config c2_h256 --no-cpu-baseline --steps 10 --warmup 3 | tail -1 | cut -c1-1800 | tee $O/c2_h256.txt
