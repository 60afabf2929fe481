#!/bin/bash
# PMC comparison of the 8-phase kernel's three operand layouts on one long-K shape (36 tiles: per-CU behaviour, no chip contention):
# LDS bank conflicts / LDS activity / wait buckets / MFMA busy.  Output: gpurun_out/pmc_p8/*.txt
R=$PWD; O=$R/gpurun_out/pmc_p8; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  for lay in "0 0" "0 1" "1 1"; do
    tag=$(echo $lay | tr -d ' ')
    timeout 120 rocprofv3 --kernel-trace --pmc $set -d $O/db_${tag}_$i -- python3 $R/tools/pmc_gemm.py 3072 768 8256 11 $lay > $O/log_${tag}_$i.txt 2>&1
    db=$(ls $O/db_${tag}_$i/*/*.db 2>/dev/null | head -1)
    echo "== layout $lay pass $i" >> $O/summary.txt
    python3 $R/tools/pmc_dump.py $db gemm_p8 >> $O/summary.txt 2>&1
    rm -rf $O/db_${tag}_$i
  done
done
cat $O/summary.txt
