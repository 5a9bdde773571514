#!/bin/bash
# usage: collect_round.sh <gpurun_out tag> <profiles prefix>  -- copy what gpu_profile_round.sh produced into profiles/ under the round's prefix
T=gpurun_out/$1; P=profiles/$2
for c in c2 c4 c4_dense c4_pixels c5 film c2_curric c2_h256 c4_h256 c4_h256_dense fwd_h256; do [ -f $T/bench_$c.json ] && cp $T/bench_$c.json ${P}_bench_$c.json; done
[ -f $T/bench_c2_profiled.json ] && cp $T/bench_c2_profiled.json ${P}_bench_c2_profiled.json
[ -f $T/bench_c2.txt ] && cp $T/bench_c2.txt ${P}_bench_c2_stdout.txt   # (round 6: the `also` lines in front of the contract line)
for c in c2 c4 c5 film c2_h256 c4_h256_dense; do [ -f $T/kernel_stats_$c.md ] && cp $T/kernel_stats_$c.md ${P}_kernel_stats_$c.md; done
for c in c2 film; do [ -s $T/step_timeline_$c.txt ] && cp $T/step_timeline_$c.txt ${P}_step_timeline_$c.txt; done
cp $T/pmc_counters.md ${P}_pmc_counters.md; cp $T/pmc_instruction_mix.md ${P}_pmc_instruction_mix.md
cp $T/pmc_traffic.json ${P}_pmc_traffic.json; cp $T/pmc_traffic.json profiles/pmc_traffic.json
cp $T/variants.txt ${P}_variants.txt; [ -f $T/ab_l0x.txt ] && cp $T/ab_l0x.txt ${P}_ab_l0x.txt
ls ${P}_* | wc -l
