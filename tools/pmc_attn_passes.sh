# PMC passes over the attention microbench (one counter set per run, kernel-trace only alongside)
R=$PWD; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_attn/p$i -o t -- python3 $R/tools/bench_attn_bwd.py c2 > $R/gpurun_out/pmc_attn_p$i.log 2>&1
  python3 $R/tools/pmc_dump.py $(ls -t $R/gpurun_out/pmc_attn/p$i/*.db $R/gpurun_out/pmc_attn/p$i/*/*.db 2>/dev/null | head -1) attn
done
