#!/bin/bash
# usage: gpu_variants.sh [--config c2|c4|c5|film] [--tu core|train_film] [--rounds N] [--cmd "<python command>"] "<flags A>" "<flags B>" ...
# Same-box A/B of kernel variants: one translation unit is rebuilt once per flag set (in parallel), every variant is linked into
# ITS OWN library (reni_amd/csrc/_build/libreni_v<i>.so, selected through RENI_HIP_LIB -- the installed library is never touched,
# so an interrupted run cannot leave a variant behind: ADVICE r02), and bench.py runs on each in turn, N times round-robin.
# "@base" as a flag set = the installed library as it is.
CFG=c2; TU=core; ROUNDS=2; CMD=""
while [[ "$1" == --* ]]; do
  case $1 in --config) CFG=$2;; --tu) TU=$2;; --rounds) ROUNDS=$2;; --cmd) CMD=$2;; esac; shift 2
done
ROOT=$(cd "$(dirname "$0")/../.."; pwd)
cd $ROOT/reni_amd/csrc; mkdir -p _build
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-spill-vgpr-to-agpr=0 -I../../include"
OBJS="core main_f32 main_bf16 film_f32 film_bf16 train_film wide shade image"
i=0; pids=()
for v in "$@"; do
  if [ "$v" != "@base" ]; then hipcc $FL $v -c reni_tu_$TU.hip -o _build/${TU}_v$i.o 2>_build/${TU}_v$i.err & pids+=($!); fi
  i=$((i+1))
  if [ $((i % 6)) = 0 ]; then for p in "${pids[@]}"; do wait $p; done; pids=(); fi
done
for p in "${pids[@]}"; do wait $p; done
i=0
for v in "$@"; do
  if [ "$v" = "@base" ]; then cp ../lib/libreni_hip.so _build/libreni_v$i.so
  else
    L=""; for o in $OBJS; do if [ $o = $TU ]; then L="$L _build/${TU}_v$i.o"; else L="$L _build/$o.o"; fi; done
    hipcc --offload-arch=gfx950 -shared -fPIC $L -o _build/libreni_v$i.so || { echo "LINK FAILED: $v"; tail -3 _build/${TU}_v$i.err; }
  fi
  i=$((i+1))
done
cd $ROOT
for round in $(seq $ROUNDS); do
  i=0
  for v in "$@"; do
    export RENI_HIP_LIB=$ROOT/reni_amd/csrc/_build/libreni_v$i.so
    if [ -n "$CMD" ]; then echo "== $v"; $CMD 2>&1 | tail -${TAILN:-4}
    else
      python bench.py --config $CFG --no-cpu-baseline --no-also --steps 30 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-44s' % '''$v''', 'Msamples/s', round(d['value']/1e6,1), ' step ms', round(d['ms_per_step'],4), ' kernel ms', round(r['kernel_avg_ms'],4), ' frac', round(r['frac'],4))"
    fi
    i=$((i+1))
  done
done
