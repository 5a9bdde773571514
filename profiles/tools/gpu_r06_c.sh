#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06c; mkdir -p $O
python profiles/tools/gpu_g14_grad_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/grad_ab.txt
timeout 1500 python -m pytest tests/test_gpu_dist.py tests/test_gpu_soak.py tests/test_gpu_train_step.py -x -q 2>&1 | tail -8 | tee $O/tests.txt
