#!/bin/bash
# usage: gpu_fwd_variants.sh "<flags A>" ... -- like gpu_variants.sh, timing the forward-only (inference) and the frozen instance
cd reni_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-spill-vgpr-to-agpr=0 -I../../include"
i=0; pids=()
for v in "$@"; do hipcc $FL $v -c reni_tu_core.hip -o _build/core_v$i.o 2>/dev/null & pids+=($!); i=$((i+1)); done
for p in "${pids[@]}"; do wait $p; done
cp ../lib/libreni_hip.so ../lib/libreni_hip.so.keep
i=0
for v in "$@"; do
  hipcc --offload-arch=gfx950 -shared -fPIC _build/core_v$i.o _build/main_f32.o _build/main_bf16.o _build/film_f32.o _build/film_bf16.o _build/train_film.o _build/shade.o _build/image.o -o ../lib/libreni_hip.so
  echo "== $v"; (cd ../..; python tests/gpu_fwd_perf.py 2>/dev/null | grep bf16; python bench.py --config c4 --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4 step ms', round(d['ms_per_step'],4), 'frozen kernel ms', round(d['roofline']['kernel_avg_ms'],4), 'stats ms', round(d['roofline'].get('stats_pass_avg_ms',0),4))")
  i=$((i+1))
done
mv ../lib/libreni_hip.so.keep ../lib/libreni_hip.so
