#!/bin/bash
# G17 (FiLM decoder training at the reference's rate) + one more L0X A/B sample in the driver's window  -> gpurun_out/r06_t_*.txt
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_trajectory.py -q -s -k "g17" > gpurun_out/r06_t_g17.txt 2>&1; echo "g17 rc=$?"
grep -E "G17|passed|failed|Error" gpurun_out/r06_t_g17.txt | head -20
bash profiles/tools/gpu_ab_driver_window.sh > gpurun_out/r06_t_ab.txt 2>&1; cat gpurun_out/r06_t_ab.txt
