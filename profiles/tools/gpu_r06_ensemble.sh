#!/bin/bash
# round 6: bench tests with the clean stdout; the G14 perturbation ensemble on the shipped bf16 kernels, on numerics ablations of them
# (RENI_ABL, reni_dev_common.inc), on the generic bf16 kernel (RENI_NO_PERSIST) and on the fp32 kernels
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_dist.py -x -q -k "bench" 2>&1 | tail -8 | tee $O/test_dist.txt
python profiles/tools/gpu_g14_ensemble.py f32 6 2>&1 | grep -v amdgpu.ids | tee $O/ens_f32.txt
RENI_NO_PERSIST=1 python profiles/tools/gpu_g14_ensemble.py bf16 6 2>&1 | grep -v amdgpu.ids | tee $O/ens_generic_bf16.txt
TAILN=9 bash profiles/tools/gpu_variants.sh --rounds 1 --cmd "python profiles/tools/gpu_g14_ensemble.py bf16 6" "@base" "-DRENI_ABL=1" "-DRENI_ABL=2" "-DRENI_ABL=3" "-DRENI_ABL=4" 2>&1 | grep -v amdgpu.ids | tee $O/ens_variants.txt
