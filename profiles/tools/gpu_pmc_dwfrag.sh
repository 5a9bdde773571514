#!/bin/bash
# wait / activity counters of k_dw_frag (config 2 at 256 features): where do its ~13 cycles per issued instruction go?
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/pmc_dwfrag; mkdir -p $O; rm -f $O/summary.md
for grp in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-50)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $O/g_$tag -o p -- python3 bench.py --config c2_h256 --steps 2 --warmup 1 --no-cpu-baseline --no-also > $O/g.log 2>&1
  python3 profiles/summarize_pmc.py $O/g_$tag/p_results.db 2>&1 | grep -i "dw_frag\|^| kernel\|^|---" >> $O/summary.md
  rm -rf $O/g_$tag
done
cat $O/summary.md | cut -c1-160
