#!/bin/bash
# round 5, final evidence that is not part of gpu_profile_round.sh: the whole GPU suite, the trajectory test's printed numbers, the cycle
# trace of the L0X training instance, timing-only ablations of k_reni_l0_ring, and k_reni_wide256 with four against eight waves
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -6 > $O/tests.log
timeout 600 python -m pytest tests/test_gpu_trajectory.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -60 > $O/trajectory.txt
TAILN=400 bash profiles/tools/gpu_variants.sh --rounds 1 --cmd "python profiles/tools/gpu_trace.py" "-DRENI_TRACE" > $O/cycle_trace.txt 2>&1
TAILN=14 bash profiles/tools/gpu_variants.sh --rounds 2 --cmd "bash profiles/tools/gpu_timeline_one.sh" "@base" "-DRENI_EXP_L0=1" "-DRENI_EXP_L0=2" "-DRENI_EXP_L0=4" 2>&1 | grep "==\|l0_ring\|step =" > $O/l0_ablation.txt
bash profiles/tools/gpu_variants.sh --tu wide --rounds 2 --cmd "python profiles/tools/gpu_c4_h256.py" "@base" "-DRENI_WIDE_WAVES=4" 2>&1 | grep "==\|H=256" > $O/wide_waves.txt
cat $O/tests.log; tail -5 $O/trajectory.txt; tail -4 $O/cycle_trace.txt; cat $O/l0_ablation.txt $O/wide_waves.txt
