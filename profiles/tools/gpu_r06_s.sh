#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "unequal_omegas" 2>&1 | grep -E "^E  |passed|failed|Error" | head -20 | cut -c1-300
