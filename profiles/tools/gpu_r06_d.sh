#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06d; mkdir -p $O
TAILN=9 bash profiles/tools/gpu_variants.sh --rounds 1 --cmd "python profiles/tools/gpu_g14_ensemble.py bf16 6" "-DRENI_ABL=8" "-DRENI_ABL=9" 2>&1 | grep -v amdgpu.ids | tee $O/ens_variants.txt
