#!/bin/bash
# GPU box: HBM fetch counter + SQ busy/wait of the split 1x1 tile GEMM at M = 12,800 / 51,200, XCD-contiguous tile order against launch order
O=/root/repo/gpurun_out/pmc_gemm_deep; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pol in ${POLICIES:-0 0x2B08580D}; do
for set in "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  tag=p${pol}_$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $O/$tag -o p -- python /root/repo/tools/bench_gemm1x1_deep.py --iters 2 --policy $pol --first 2 > $O/$tag.log 2>&1
  DB=$(find $O/$tag -name "*.db" | head -1)
  echo "== policy $pol: $set"
  [ -n "$DB" ] && python /root/repo/tools/rocpd_pmc.py $DB gemm1x1 2>&1 | tail -8
  find $O/$tag -name "*.db" -delete
done
done
