#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
for r in 1 2; do
  for p in 0x580D 0x100580D; do
    python tools/profile_conv_shapes.py --no-overlap --policy $p 2>/dev/null | grep "k=3x3" | sed "s/^/r$r $p /" >> $O/r3_k32_shapes.txt
  done
done
python - <<'PY'
import collections, re
d = collections.defaultdict(lambda: collections.defaultdict(list))
for line in open('gpurun_out/r3_k32_shapes.txt'):
    f = line.split()
    pol = f[1]; key = ' '.join(f[2:5]); us = float(line.split('split-bf16')[1].split()[2])
    d[key][pol].append(us)
for k, v in d.items():
    a = sum(v['0x580D'])/len(v['0x580D']); b = sum(v['0x100580D'])/len(v['0x100580D'])
    print(f"{k:40s} 32x32x16 {a:7.1f} us   16x16x32 {b:7.1f} us   {100*(b/a-1):+5.1f} %")
PY
