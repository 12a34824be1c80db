#!/bin/bash
# Round 6, call F: split-operand 1x1 GEMMs from M = 6,400: parity (GEMM tests + full-size goldens), configs[1] / configs[2] bench.
cd "$(dirname "$0")/.."
O=gpurun_out/r6f; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "gemm or conv or linear or ln_" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 900 python -m pytest tests/test_hip_fullsize.py tests/test_hip_end2end.py tests/test_hip_ctx.py -x -q > $O/pytest_full.log 2>&1; echo "pytest fullsize rc=$?"; tail -3 $O/pytest_full.log
for rep in 1 2; do
v=$(timeout 400 python bench.py --res 128 --frames 400 --no-cpu-baseline --no-max-clip --no-decode --no-kernel-events --no-shard-sim --no-other-configs --steps 2 --warmup 1 2>/dev/null | tail -1 |
    python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['value'], 2), round(d['ms_per_step'], 1))")
echo "configs[1] 128px 400f: $v" | tee -a $O/bench.txt
v=$(timeout 400 python bench.py --no-cpu-baseline --no-max-clip --no-decode --no-kernel-events --no-shard-sim --no-other-configs --steps 2 --warmup 1 2>/dev/null | tail -1 |
    python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['value'], 2), round(d['ms_per_step'], 1))")
echo "configs[2] 256px 200f: $v" | tee -a $O/bench.txt
done
timeout 300 python tools/profile_conv_shapes.py --frames 400 --res 128 2>&1 | grep -v amdgpu > $O/insitu_c1.txt; head -24 $O/insitu_c1.txt
