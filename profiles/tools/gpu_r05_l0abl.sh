#!/bin/bash
# timing-only ablations of k_reni_l0_ring (results wrong by construction): rocprof timeline of one step per variant
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAILN=${TAILN:-14} bash profiles/tools/gpu_variants.sh --rounds 1 --cmd "bash profiles/tools/gpu_timeline_one.sh" "$@" 2>&1 | grep "==\|l0_ring"
