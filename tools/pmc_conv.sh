#!/bin/bash
# GPU box: PMC passes for the conv_gemm microbenchmark (one --pmc set per run; FETCH/WRITE in their own passes).
# usage: tools/pmc_conv.sh <case> <outdir> [variants] [sets]
CASE=$1; OUT=$2; VAR=${3:-7}; SETS=${4:-"sq1 sq2 tcc1 tcc2 tcc3"}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; timeout 200 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/$tag -o p -- python /root/repo/tools/bench_conv.py --cases $CASE --iters 3 --variants $VAR > $OUT/$tag.log 2>&1; }
for s in $SETS; do
case $s in
sq1) run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES;;
sq2) run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE;;
tcc1) run tcc1 FETCH_SIZE;;
tcc2) run tcc2 WRITE_SIZE;;
tcc3) run tcc3 TCC_HIT_sum TCC_MISS_sum;;
esac
done
