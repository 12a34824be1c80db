#!/bin/bash
# GPU box: SQ counters of the fused 64-channel temporal layer, 32 x 32 kernel (WMODE 3) vs window-tiled (WMODE 4), at 200 frames x 4096 pixels
# (tools/bench_temporal_layer.py), two --pmc passes (8 SQ slots each), kernel-trace only.   bash tools/pmc_temporal_layer.sh [outdir]
OUT=${1:-/root/repo/gpurun_out/pmc_temporal}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace \
  -d $OUT/run1 -o p -- python /root/repo/tools/bench_temporal_layer.py > $OUT/run1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM --kernel-trace \
  -d $OUT/run2 -o p -- python /root/repo/tools/bench_temporal_layer.py > $OUT/run2.log 2>&1
cd /root/repo
python tools/pmc_temporal_post.py "$OUT" | tee "$OUT/pmc_temporal_layer.md"
