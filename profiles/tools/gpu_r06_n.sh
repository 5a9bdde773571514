#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06n; mkdir -p $O
TAILN=1 bash profiles/tools/gpu_variants.sh --rounds 2 --cmd "timeout 120 python profiles/tools/gpu_frozen_time.py" "@base" "-DRENI_ALT=1" 2>&1 | grep -v amdgpu.ids | tee $O/frozen_alt.txt
