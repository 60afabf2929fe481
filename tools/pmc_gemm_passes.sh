R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 60 rocprofv3 --list-avail > $R/gpurun_out/pmc_avail.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM"; do
  i=$((i+1))
  for shp in "4096 4096 4096" "8256 3072 768"; do
    tag=$(echo $shp | tr ' ' 'x')
    timeout 120 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_g/$tag/p$i -- python3 $R/tools/pmc_gemm.py $shp 3 > $R/gpurun_out/pmc_g_${tag}_p$i.log 2>&1
  done
done
echo ok
