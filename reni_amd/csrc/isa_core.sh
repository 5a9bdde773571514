#!/bin/bash
# emit the gfx950 ISA of the core translation unit (build.sh's flags) to $1 and print the training kernel's spill / register summary
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-spill-vgpr-to-agpr=0 -I../../include "${@:2}" -S --cuda-device-only reni_tu_core.hip -o "$1" 2>&1 | grep -E "error" 
awk '/^_ZN4reni17k_reni_train_bf16ILi128ELb1ELb0EEEvNS_8MainArgsE:/{p=1} p&&/scratch_/{print NR": "$0} p&&/^\.Lfunc_end0/{exit}' "$1" | head
grep -n "; ScratchSize\|; NumVgprs" "$1" | head -6
