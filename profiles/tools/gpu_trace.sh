#!/bin/bash
# usage: gpu_trace.sh [extra hipcc flags]  -- s_memtime trace of one workgroup's tile loop (tags: TRACE(n) in k_reni_train_bf16)
ROOT=$(cd "$(dirname "$0")/../.."; pwd)
profiles/tools/gpu_variants.sh --cmd "python profiles/tools/gpu_trace.py" "-DRENI_TRACE $*" | tail -400
