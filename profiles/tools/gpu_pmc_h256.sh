#!/bin/bash
# usage: gpu_pmc_h256.sh <out.md>  -- instruction mix / wait counters of the H = 256 chain (k_reni_main<bf16,256,FWD_BWD>) and k_dw_frag
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=$1; rm -f $OUT
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVES"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/h_$tag -o p -- python3 bench.py --config c2_h256 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python3 profiles/summarize_pmc.py gpurun_out/h_$tag/p_results.db 2>&1 | grep -i "k_reni_main\|k_dw_frag\|^| kernel\|^|---" >> $OUT
  rm -rf gpurun_out/h_$tag
done
cat $OUT
