#!/bin/bash
# GPU box: SQ wait / busy counters of the row-stationary GEMM (one --pmc set per pass, kernel-trace only)
O=/root/repo/gpurun_out/pmc_gemm; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C="python /root/repo/tools/bench_gemm1x1.py --policy 0x580D --iters 3 --first 1"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $O/$tag -o p -- $C > $O/$tag.log 2>&1
  DB=$(find $O/$tag -name "*.db" | head -1)
  [ -n "$DB" ] && python /root/repo/tools/rocpd_pmc.py $DB rowreg 2>&1 | tail -12
  find $O/$tag -name "*.db" -delete
done
