#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06m; mkdir -p $O
TAILN=1 bash profiles/tools/gpu_variants.sh --rounds 3 --cmd "python profiles/tools/gpu_frozen_time.py" "@base" "-DRENI_PRIO=1" "-DRENI_PRIO=2" 2>&1 | grep -v amdgpu.ids | tee $O/frozen_prio.txt
bash profiles/tools/gpu_ab_driver_window.sh 2>&1 | tee $O/ab_l0x_driver_window.txt
