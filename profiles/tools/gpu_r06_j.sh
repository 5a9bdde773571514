#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06j; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "cosine_loss_on_the_persistent" 2>&1 | grep -v amdgpu.ids | tail -40 | cut -c1-600 | tee $O/cos.txt
RENI_HIP_LIB=$PWD/reni_amd/csrc/_build/libreni_r06_before_consistent_training.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "cosine_loss_on_the_persistent" 2>&1 | tail -3 | tee $O/cos_before.txt
