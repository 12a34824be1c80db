#!/bin/bash
# GPU box: SQ counters of the row-stationary GEMM at its largest shape (M = 819,200, N = 64, K = 128, + residual): one --pmc set per pass
O=/root/repo/gpurun_out/pmc_rowreg; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C="python /root/repo/tools/bench_gemm1x1.py --iters 3 --only ${1:-9}"
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $O/$tag -o p -- $C > $O/$tag.log 2>&1
  DB=$(find $O/$tag -name "*.db" | head -1)
  [ -n "$DB" ] && python /root/repo/tools/rocpd_pmc.py $DB gemm1x1 2>&1 | tail -6
  find $O/$tag -name "*.db" -delete
done
