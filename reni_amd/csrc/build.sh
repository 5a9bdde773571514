#!/bin/bash
# Build libreni_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
# The translation units reni_tu_*.hip are compiled in parallel (see the header of reni_device.inc).
set -e
cd "$(dirname "$0")"
mkdir -p ../lib _build
# -amdgpu-spill-vgpr-to-agpr=0: the training kernel owns the AGPRs by hand (see mfma_bf16_agpr_tile)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-spill-vgpr-to-agpr=0 -I../../include"
pids=()
for tu in core main_f32 main_bf16 film_f32 film_bf16 train_film wide shade image; do
  hipcc $FLAGS "$@" -c reni_tu_$tu.hip -o _build/$tu.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC _build/core.o _build/main_f32.o _build/main_bf16.o _build/film_f32.o _build/film_bf16.o _build/train_film.o _build/wide.o _build/shade.o _build/image.o -o ../lib/libreni_hip.so
