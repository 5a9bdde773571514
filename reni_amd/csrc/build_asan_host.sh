#!/bin/bash
# Host-only AddressSanitizer build of the C ABI (CPU build only -- never for the GPU): the library's translation units compiled with
# --cuda-host-only (no device code, seconds), linked with tests/capi/capi_args.c into _build/asan/capi_args.  Run by
# tests/test_capi_asan_cpu.py.  usage: build_asan_host.sh
set -e
cd "$(dirname "$0")"
mkdir -p _build/asan
FLAGS="--offload-arch=gfx950 --cuda-host-only -O1 -g -std=c++17 -fPIC -fsanitize=address -fno-omit-frame-pointer -I../../include"
pids=()
rm -f _build/asan/*.o _build/asan/*.stub
for tu in core main_f32 main_bf16 film_f32 film_bf16 train_film wide shade image; do
  hipcc $FLAGS -c reni_tu_$tu.hip -o _build/asan/$tu.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
# the host objects refer to their (absent) device code as __hip_fatbin_<hash>: an EMPTY offload bundle stands in for each (the HIP
# runtime registers the pointer at load and opens it only when a kernel is launched -- which this binary never does)
{ for o in _build/asan/*.o; do nm -u $o; done; } | grep -o "__hip_fatbin_[0-9a-f]*" | sort -u | \
  awk '{ printf "const char %s[32] __attribute__((aligned(4096))) = \"__CLANG_OFFLOAD_BUNDLE__\";\n", $1 }' > _build/asan/fatbin_stubs.c
/opt/rocm/lib/llvm/bin/clang -c _build/asan/fatbin_stubs.c -o _build/asan/fatbin_stubs.o.stub
/opt/rocm/lib/llvm/bin/clang -O1 -g -fsanitize=address -fno-omit-frame-pointer -I../../include -c ../../tests/capi/capi_args.c -o _build/asan/capi_args.o
hipcc -fsanitize=address _build/asan/capi_args.o _build/asan/fatbin_stubs.o.stub _build/asan/core.o _build/asan/main_f32.o _build/asan/main_bf16.o _build/asan/film_f32.o \
  _build/asan/film_bf16.o _build/asan/train_film.o _build/asan/wide.o _build/asan/shade.o _build/asan/image.o -o _build/asan/capi_args
