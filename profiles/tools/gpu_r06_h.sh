#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06h; mkdir -p $O
TAILN=1 bash profiles/tools/gpu_variants.sh --rounds 2 --cmd "python profiles/tools/gpu_frozen_time.py" "@base" "-DRENI_EXP=64" "-DRENI_EXP=192" "-DRENI_EXP=1" "-DRENI_EXP=8" "-DRENI_EXP=1024" "-DRENI_EXP=32768" "-DRENI_EXP=33792" "-DRENI_EXP=34017" "-DRENI_EXP=193" 2>&1 | grep -v amdgpu.ids | tee $O/frozen_ablation2.txt
