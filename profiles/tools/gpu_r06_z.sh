#!/bin/bash
# the records of FINDING 3 (profiles/r06_trajectory.md section 8) -> gpurun_out/r06_z_*.txt
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{ echo "== python profiles/tools/gpu_g20_ensemble.py bf16 6   (shipped: k_reni_wide256<1>)"; python profiles/tools/gpu_g20_ensemble.py bf16 6 2>&1 | tail -9
  echo "== RENI_NO_PERSIST=1 python profiles/tools/gpu_g20_ensemble.py bf16 6   (generic bf16 kernels)"; RENI_NO_PERSIST=1 python profiles/tools/gpu_g20_ensemble.py bf16 6 2>&1 | tail -9; } > gpurun_out/r06_z_ensemble.txt
{ echo "== python profiles/tools/gpu_g20_consistency.py 256"; python profiles/tools/gpu_g20_consistency.py 256 2>&1 | grep -v amdgpu.ids
  echo "== python profiles/tools/gpu_g20_consistency.py 128"; python profiles/tools/gpu_g20_consistency.py 128 2>&1 | grep -v amdgpu.ids
  echo "== python profiles/tools/gpu_g20_along.py"; python profiles/tools/gpu_g20_along.py 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r06_z_consistency.txt
{ echo "== python profiles/tools/gpu_g20_forward.py 256 / 128"; python profiles/tools/gpu_g20_forward.py 256 2>&1 | grep width; python profiles/tools/gpu_g20_forward.py 128 2>&1 | grep width
  echo "== python profiles/tools/gpu_g20_emu_check.py 256"; python profiles/tools/gpu_g20_emu_check.py 2>&1 | grep -v amdgpu.ids
  echo "== python profiles/tools/gpu_g20_emu_check.py 128"; python profiles/tools/gpu_g20_emu_check.py 128 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r06_z_forward_emu.txt
tail -3 gpurun_out/r06_z_consistency.txt
