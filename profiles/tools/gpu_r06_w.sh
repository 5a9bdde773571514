#!/bin/bash
# after the idle-group fix of k_reni_wide256<2, FILM>: the FiLM tests, the 600-case fuzz, then 1200 cases from another seed window
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_film.py tests/test_gpu_wide_train.py -q -x -p no:cacheprovider > gpurun_out/r06_w_film.txt 2>&1; echo "film rc=$?"; tail -3 gpurun_out/r06_w_film.txt
RENI_FUZZ_CASES=1500 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q --timeout 2300 -p no:cacheprovider > gpurun_out/r06_w_fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -8 gpurun_out/r06_w_fuzz.txt | cut -c1-300
