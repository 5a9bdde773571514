#!/bin/bash
# usage: gpu_profile_round.sh <tag>  -- bench lines (c2 with its sub-records, c4, c5, film), rocprofv3 kernel stats + step timeline of
# the same bench command, HBM-traffic PMC passes for every dominant kernel (separate --pmc runs, no trace domains), instruction-mix /
# wait counters, and the variants table.  Everything lands under gpurun_out/<tag>/; copy what is to be judged into profiles/.
TAG=$1
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
rm -f $O/pmc_counters.md $O/pmc_instruction_mix.md
for c in c2 c4 c5 film c2_h256; do
  for cnt in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $cnt --kernel-trace -d $O/pmc_$cnt -o p -- python3 bench.py --config $c --steps 3 --warmup 2 --no-cpu-baseline --no-also > $O/pmc_$cnt.log 2>&1
    python3 profiles/summarize_pmc.py $O/pmc_$cnt/p_results.db >> $O/pmc_counters.md 2>&1
    rm -rf $O/pmc_$cnt
  done
done
# (round 5) the H = 256 persistent chain on its own: config 4's step (frozen decoder, every tile) and config 5's shape forward
for x in "c4 --hidden 256 --dense" "c5 --hidden 256"; do
  for cnt in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $cnt --kernel-trace -d $O/pmc_$cnt -o p -- python3 bench.py --config $x --steps 3 --warmup 2 --no-cpu-baseline --no-also > $O/pmc_$cnt.log 2>&1
    python3 profiles/summarize_pmc.py $O/pmc_$cnt/p_results.db >> $O/pmc_counters.md 2>&1
    rm -rf $O/pmc_$cnt
  done
done
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_MFMA" "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-60)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $O/g_$tag -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also > $O/g_$tag.log 2>&1
  python3 profiles/summarize_pmc.py $O/g_$tag/p_results.db 2>&1 | grep -i "train_bf16\|dw1\|l0_ring\|wide256\|^##\|^| kernel\|^|---" >> $O/pmc_instruction_mix.md
  rm -rf $O/g_$tag $O/g_$tag.log
done
python3 profiles/make_pmc_traffic.py $O/pmc_counters.md $O/pmc_instruction_mix.md > $O/pmc_traffic.json 2>> $O/bench.err
cp $O/pmc_traffic.json profiles/pmc_traffic.json   # (so that the bench lines below carry the traffic of THIS source state)
# (round 6: bench.py prints `also <name> {json}` lines and then ONE contract line: the whole stdout is kept as .txt, the line as .json)
run_bench() { out=$1; shift; python bench.py "$@" > $O/$out.txt 2>> $O/bench.err; tail -n 1 $O/$out.txt > $O/$out.json; cut -c1-200 $O/$out.json; }
for c in c2 c4 c5 film c2_curric c2_h256; do
  X="--config $c --no-cpu-baseline"; [ $c = c2 ] && X="--steps 20 --warmup 5"   # (c2: exactly the driver's command)
  run_bench bench_$c $X
done
run_bench bench_c4_dense --config c4 --dense --no-cpu-baseline        # RENI_WEIGHT_SPARSE off
run_bench bench_c4_pixels --config c4 --pixels --no-cpu-baseline      # RENI_WEIGHT_COMPACT
run_bench bench_c4_h256 --config c4 --hidden 256 --no-cpu-baseline
run_bench bench_c4_h256_dense --config c4 --hidden 256 --dense --no-cpu-baseline
run_bench bench_fwd_h256 --config c5 --hidden 256 --no-cpu-baseline
for c in c2 c4 c5 film c2_h256; do
  # c2: the DEFAULT command as the driver runs it (the headline's 5 + 20 steps first, the sub-records, the same steps again): the
  # summary's last two lines are the averages of the headline's and of the sustained window's 20 timed launches
  X="--config $c --steps 10 --warmup 2 --no-also"; K=""; [ $c = c2 ] && { X="--steps 20 --warmup 5"; K="20 5"; }
  rocprofv3 --kernel-trace --stats -d $O/kt_$c -o k -- python3 bench.py $X --no-cpu-baseline > $O/kt_$c.txt 2> $O/kt_$c.log
  [ $c = c2 ] && tail -n 1 $O/kt_$c.txt > $O/bench_c2_profiled.json   # (the line of the PROFILED process: its kernel_avg_ms is what the summary's headline window must agree with)
  python3 profiles/summarize_rocpd.py $O/kt_$c/k_results.db $O/kernel_stats_$c.md $K > /dev/null 2>&1 || ls -R $O/kt_$c | head
  { [ $c = c2 ] || [ $c = film ] || [ $c = c2_h256 ]; } && python3 profiles/timeline_rocpd.py $O/kt_$c/k_results.db > $O/step_timeline_$c.txt 2>/dev/null
  rm -rf $O/kt_$c
done
rocprofv3 --kernel-trace --stats -d $O/kt_c4h -o k -- python3 bench.py --config c4 --hidden 256 --dense --steps 10 --warmup 2 --no-also --no-cpu-baseline > $O/kt_c4h.log 2>&1
python3 profiles/summarize_rocpd.py $O/kt_c4h/k_results.db $O/kernel_stats_c4_h256_dense.md > /dev/null 2>&1; rm -rf $O/kt_c4h
# the L0X split against round 4's kernels on THIS box (same library, RENI_NO_L0X read at plan creation): alternating, 3 rounds
bash profiles/tools/gpu_ab_env.sh 3 "-" "RENI_NO_L0X=1" > $O/ab_l0x.txt 2>&1
python profiles/tools/gpu_perf_variants.py > $O/variants.txt 2>&1
head -8 $O/kernel_stats_c2.md; cat $O/pmc_traffic.json; tail -12 $O/variants.txt
