#!/bin/bash
# Round 6, call E: 3x3 convs on 4 x 4-pixel frames through the v2 kernel's 36-segment instantiations: parity, configs[1] bench.
cd "$(dirname "$0")/.."
O=gpurun_out/r6e; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "conv_gemm or fused_gn_stats or conv_bf16_split" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 600 python -m pytest tests/test_hip_fullsize.py -x -q > $O/pytest_full.log 2>&1; echo "pytest fullsize rc=$?"; tail -3 $O/pytest_full.log
for rep in 1 2; do
v=$(timeout 400 python bench.py --res 128 --frames 400 --no-cpu-baseline --no-max-clip --no-decode --no-kernel-events --no-shard-sim --no-other-configs --steps 2 --warmup 1 2>/dev/null | tail -1 |
    python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['value'], 2), round(d['ms_per_step'], 1))")
echo "configs[1] 128px 400f: $v" | tee -a $O/bench_c1.txt
done
timeout 300 python tools/profile_conv_shapes.py --frames 400 --res 128 2>&1 | grep -v amdgpu | head -14 | tee $O/insitu_c1.txt
