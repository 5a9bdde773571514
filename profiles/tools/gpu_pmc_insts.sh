cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_g
rocprofv3 --list-avail > gpurun_out/pmc_g/avail.txt 2>&1
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_WAVE32_LDS"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-60)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/pmc_g/$tag -o p -- python3 bench.py --steps 3 --warmup 1 > gpurun_out/pmc_g/$tag.log 2>&1
  python3 profiles/summarize_pmc.py gpurun_out/pmc_g/$tag/p_results.db 2>&1 | grep -i "train_bf16\|dw1\|^##" >> gpurun_out/pmc_g/summary.md
  rm -rf gpurun_out/pmc_g/$tag
done
cat gpurun_out/pmc_g/summary.md
