#!/bin/bash
# the many-tiles fuzz (tests/test_gpu_fuzz.py::test_fuzz_many_tiles_against_oracle) at N cases
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
RENI_FUZZ_MEDIUM=${1:-600} RENI_FUZZ_CASES=1 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -q -k many_tiles --timeout 2900 -p no:cacheprovider > gpurun_out/r06_x_fuzz.txt 2>&1; echo "fuzz rc=$?"
grep -n "^FAILED\|passed\|failed" gpurun_out/r06_x_fuzz.txt | cut -c1-250 | tail -40
