#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06e; mkdir -p $O
python profiles/tools/gpu_g14_ensemble.py bf16 6 2>&1 | grep -v amdgpu.ids | tee $O/ens_bf16_consistent.txt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/tests.txt
