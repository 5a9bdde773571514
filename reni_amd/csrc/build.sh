#!/bin/bash
# Build libreni_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
mkdir -p ../lib
# -amdgpu-spill-vgpr-to-agpr=0: the training kernel owns the AGPRs by hand (see mfma_bf16_agpr_tile)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-spill-vgpr-to-agpr=0 -I../../include reni_kernels.hip -o ../lib/libreni_hip.so "$@"
