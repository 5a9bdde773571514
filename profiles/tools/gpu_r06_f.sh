#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06f; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_trajectory.py -x -q -s 2>&1 | grep -v amdgpu.ids | grep "G14\|G15\|passed\|failed\|Error\|assert" | cut -c1-400 | tee $O/trajectory.txt
TAILN=6 bash profiles/tools/gpu_variants.sh --rounds 1 --cmd "python -m pytest tests/test_gpu_trajectory.py -q -s -k bf16_g15" "@base" "-DRENI_ABL=32" 2>&1 | grep -v amdgpu.ids | grep "==\|G15 bf16: max\|G15 bf16: final\|passed\|failed" | tee $O/g15_consistent_training.txt
bash profiles/tools/gpu_ab_driver_window.sh 2>&1 | tee $O/ab_l0x_driver_window.txt
python bench.py --config c4 --no-cpu-baseline --steps 50 --warmup 20 | tail -1 | cut -c1-1500 | tee $O/c4.txt
python bench.py --config c4 --dense --no-cpu-baseline --steps 50 --warmup 20 | tail -1 | cut -c1-1500 | tee $O/c4_dense.txt
python bench.py --gpus 1 --comm capi --no-also --no-cpu-baseline | tail -1 | cut -c1-2500 | tee $O/capi.txt
