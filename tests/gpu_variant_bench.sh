#!/bin/bash
# usage: gpu_variant_bench.sh "<flags A>" "<flags B>" ...   -- rebuilds the core TU with each flag set and runs bench.py
cd reni_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-spill-vgpr-to-agpr=0 -I../../include"
for v in "$@"; do
  if [ "$v" = "@prev" ]; then cp _build/core_prev.o _build/core.o   # a core TU built beforehand from another revision
  else hipcc $FL $v -c reni_tu_core.hip -o _build/core.o 2>/dev/null; fi
  hipcc --offload-arch=gfx950 -shared -fPIC _build/core.o _build/main_f32.o _build/main_bf16.o _build/film_f32.o _build/film_bf16.o _build/train_film.o _build/shade.o _build/image.o -o ../lib/libreni_hip.so
  for i in 1 2 3; do
    (cd ../..; python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,1), round(d['ms_per_step'],4), round(d['roofline']['kernel_avg_ms'],4))")
  done
done
