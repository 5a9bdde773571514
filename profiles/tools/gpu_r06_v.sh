#!/bin/bash
# wide fuzz on the final sources: 600 random problems against the oracle, 1500 random masks sparse / compact against dense
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
RENI_FUZZ_CASES=600 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -x --timeout 1400 -p no:cacheprovider > gpurun_out/r06_v_fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -5 gpurun_out/r06_v_fuzz.txt
timeout 900 python profiles/tools/gpu_sparse_fuzz.py 1500 7 > gpurun_out/r06_v_sparse.txt 2>&1; echo "sparse rc=$?"; tail -4 gpurun_out/r06_v_sparse.txt
