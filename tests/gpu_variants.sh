#!/bin/bash
# usage: gpu_variants.sh "<flags A>" "<flags B>" ...  -- builds the core TU once per flag set (in parallel), then runs bench.py on
# each library in turn, twice round-robin (same box, same call: the only comparisons that can be trusted to +-0.5 %)
cd reni_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-spill-vgpr-to-agpr=0 -I../../include"
i=0; pids=()
for v in "$@"; do
  if [ "$v" = "@prev" ]; then cp _build/core_prev.o _build/core_v$i.o &   # a core TU built beforehand from another revision
  else hipcc $FL $v -c reni_tu_core.hip -o _build/core_v$i.o 2>/dev/null & fi
  pids+=($!); i=$((i+1))
  if [ $((i % 6)) = 0 ]; then for p in "${pids[@]}"; do wait $p; done; pids=(); fi
done
for p in "${pids[@]}"; do wait $p; done
cp ../lib/libreni_hip.so ../lib/libreni_hip.so.keep
for round in 1 2; do
  i=0
  for v in "$@"; do
    hipcc --offload-arch=gfx950 -shared -fPIC _build/core_v$i.o _build/main_f32.o _build/main_bf16.o _build/film_f32.o _build/film_bf16.o _build/train_film.o _build/shade.o _build/image.o -o ../lib/libreni_hip.so
    (cd ../..; python bench.py --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-44s' % '$v', 'Msamples/s', round(d['value']/1e6,1), ' step ms', round(d['ms_per_step'],4), ' kernel ms', round(d['roofline']['kernel_avg_ms'],4))")
    i=$((i+1))
  done
done
mv ../lib/libreni_hip.so.keep ../lib/libreni_hip.so
