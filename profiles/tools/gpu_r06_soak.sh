#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06soak; mkdir -p $O
timeout 900 python profiles/tools/gpu_soak.py 2>&1 | grep -v amdgpu.ids | tee $O/soak.txt
bash profiles/tools/gpu_ab_driver_window.sh 2>&1 | tee $O/ab_l0x_driver_window.txt
