#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06o; mkdir -p $O
python profiles/tools/gpu_film_wide_check.py 2>&1 | grep -v amdgpu.ids | tee $O/film_wide_check.txt
python profiles/tools/gpu_film_fwd_h256.py 2>&1 | grep -v amdgpu.ids | tee $O/film_fwd_after.txt
