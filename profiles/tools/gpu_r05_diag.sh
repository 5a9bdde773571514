#!/bin/bash
# round 5 diagnostics of the L0X split: step timeline, cycle trace of the L0X training instance, cycle trace of k_reni_l0_ring
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_diag; mkdir -p $O
bash profiles/tools/gpu_variants.sh --rounds 1 --cmd "bash profiles/tools/gpu_timeline_one.sh" "@base" > $O/timeline.txt 2>&1
TAILN=400 bash profiles/tools/gpu_variants.sh --rounds 1 --cmd "python profiles/tools/gpu_trace.py" "-DRENI_TRACE" > $O/trace_train.txt 2>&1
TAILN=500 bash profiles/tools/gpu_variants.sh --rounds 1 --cmd "python profiles/tools/gpu_trace_dw1.py" "-DRENI_TRACE_DW1 -DRENI_L0_NSLOT=3" > $O/trace_l0ring.txt 2>&1
tail -20 $O/timeline.txt
