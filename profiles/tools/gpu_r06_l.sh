#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06l; mkdir -p $O
TAILN=120 bash profiles/tools/gpu_variants.sh --rounds 1 --cmd "python profiles/tools/gpu_trace_frozen.py" "-DRENI_TRACE" "-DRENI_TRACE -DRENI_TRACE_WAVE=4" 2>&1 | grep -v amdgpu.ids | tee $O/trace_frozen.txt
bash profiles/tools/gpu_ab_driver_window.sh 2>&1 | tee $O/ab_l0x_driver_window.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
