#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06p; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_film.py -x -q 2>&1 | tail -12 | cut -c1-300 | tee $O/film_tests.txt
python profiles/tools/gpu_film_train_h256.py 2>&1 | grep -v amdgpu.ids | tee $O/film_train.txt
RENI_NO_PERSIST=1 python profiles/tools/gpu_film_train_h256.py 2>&1 | grep -v amdgpu.ids | tee -a $O/film_train.txt
