#!/usr/bin/env python3
"""Turn the error logs written by `pytest -m gpu` (gpurun_out/op_errors.jsonl, e2e_errors.jsonl) into
profiles/<name>.md:   python tools/parity_table.py gpurun_out profiles/r1_parity_errors.md"""
import json, os, sys

src, dst = sys.argv[1], sys.argv[2]
ops, e2e, extra = {}, {}, []
for line in open(os.path.join(src, "op_errors.jsonl")):
    r = json.loads(line)
    if "max_abs_err" in r:
        ops[r["op"]] = r
    else:
        extra.append(r)
p = os.path.join(src, "e2e_errors.jsonl")
if os.path.exists(p):
    for line in open(p):
        r = json.loads(line)
        e2e[r["case"]] = r
with open(dst, "w") as f:
    f.write("# Parity errors measured on MI355X (gpurun, `pytest -m gpu`, current build), HIP vs the CPU oracle / torch fp32 op references\n\n")
    f.write("## End to end (HIP path vs `oracle/dawn_oracle.py` and the reference-generated goldens)\n\n| case | max abs err | max |ref| |\n|---|---|---|\n")
    for k, r in e2e.items():
        f.write(f"| {k} | {r['max_abs_err']:.3e} | {r['max_abs_ref']:.3g} |\n")
    if extra:
        f.write("\n## Accuracy against fp64 references (split-operand convolution: relative max error; flow decode: max abs error of the HIP path and of the fp32 CPU oracle)\n")
        last = {}
        for r in extra:
            last[r["op"]] = r                      # latest record per check
        for r in last.values():
            f.write(f"\n`{r['op']}`\n\n")
            f.write("| " + " | ".join(k for k in r if k != "op") + " |\n|" + "---|" * (len(r) - 1) + "\n")
            f.write("| " + " | ".join(f"{v:.3e}" for k, v in r.items() if k != "op") + " |\n")
    f.write("\n## Per kernel / op case\n\n| op / case | max abs err | scale (max|ref|) | tolerance |\n|---|---|---|---|\n")
    for k, r in ops.items():
        f.write(f"| {k} | {r['max_abs_err']:.3e} | {r['scale']:.3g} | {r['tol']:.1e} |\n")
print(f"{len(e2e)} end-to-end cases, {len(ops)} op cases -> {dst}")
