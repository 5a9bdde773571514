#!/bin/bash
# the many-tiles fuzz (300 cases) under each path selector: every alternative path is a shipped, selectable path
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for e in "RENI_FRAG_WS_CAP_MB=8" "RENI_NO_SIDE_STREAM=1" "RENI_NO_L0X=1" "RENI_DW1_OLD=1" "RENI_NO_PERSIST=1"; do
  echo "== $e"
  env $e RENI_FUZZ_MEDIUM=300 RENI_FUZZ_CASES=1 RENI_FUZZ_FORWARD=1 RENI_FUZZ_LOSSES=1 RENI_FUZZ_ENGINE=1 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -k many_tiles --timeout 580 -p no:cacheprovider 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-250
done
