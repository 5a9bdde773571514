#!/bin/bash
# usage: gpu_variants_cfg.sh <bench config> "<flags A>" "<flags B>" ...  -- as gpu_variants.sh, for another bench configuration
CFG=$1; shift
cd reni_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-spill-vgpr-to-agpr=0 -I../../include"
i=0; pids=()
for v in "$@"; do hipcc $FL $v -c reni_tu_core.hip -o _build/core_v$i.o 2>/dev/null & pids+=($!); i=$((i+1)); done
for p in "${pids[@]}"; do wait $p; done
cp ../lib/libreni_hip.so ../lib/libreni_hip.so.keep
for round in 1 2 3; do
  i=0
  for v in "$@"; do
    hipcc --offload-arch=gfx950 -shared -fPIC _build/core_v$i.o _build/main_f32.o _build/main_bf16.o _build/film_f32.o _build/film_bf16.o _build/train_film.o _build/shade.o _build/image.o -o ../lib/libreni_hip.so
    (cd ../..; python bench.py --config $CFG --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-30s' % '$v', 'Msamples/s', round(d['value']/1e6,1), ' step ms', round(d['ms_per_step'],4), ' kernel ms', round(d['roofline']['kernel_avg_ms'],4))")
    i=$((i+1))
  done
done
mv ../lib/libreni_hip.so.keep ../lib/libreni_hip.so
