#!/bin/bash
cd $GRAFT_REPO_ROOT
python profiles/tools/gpu_film_wide_diag.py 2>&1 | grep -v amdgpu.ids | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_film.py -q 2>&1 | grep -E "^E  |passed|failed" | head -20 | cut -c1-300
python profiles/tools/gpu_film_train_h256.py 2>&1 | grep "FiLM"
