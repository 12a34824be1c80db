#!/usr/bin/env python3
"""Static instruction breakdown of conv3x3_wino_kernel (VERDICT r4 #1a: where do the VALU instructions of the Winograd launch go?).

Compiles dawn-pytorch_amd/csrc/conv3x3_wino.hip to gfx950 assembly (hipcc cross-compiles: no GPU needed), finds the chunk loop
(the depth-2 loop of the persistent tile loop), cuts it at its two barriers into phase A / phase B, takes the rest of the tile
loop as the epilogue (+ per-tile bookkeeping), and classifies every instruction:

  mfma        v_mfma_*
  split       the exact 3-way bf16 split of the transformed values: v_and_b32 with 0xffff0000, the residual v_sub_f32 that follow,
              v_perm_b32 (packing)              [8 v_and + 8 v_sub + 6 v_perm per 4 values, dawn_common.h]
  transform   B^T d B: the remaining v_add_f32 / v_sub_f32 / v_fma_f32 / v_mul_f32 / v_pk_* on fp32 data
  out-xform   (epilogue only) A^T M A adds, bias / residual adds, GroupNorm sums -- every fp32 VALU op of the epilogue
  address     integer VALU: v_add_u32, v_lshl*, v_and/or on indices, v_mad_u*, v_mov, v_cndmask, v_readlane, ...
  lds / vmem / salu / wait / barrier / branch

and multiplies by the dynamic trip counts of a launch shape (tiles x chunks) to predict SQ_INSTS_VALU, which the PMC pass of
tools/pmc_wino_shapes.sh measures (profiles/r5_wino_valu_breakdown.md holds both).

    python tools/isa_breakdown.py [--md out.md]
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_asm(src, extra=()):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", *extra, src, "-o", out],
                          stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def kernel_lines(lines, marker):
    """Instruction lines of the first kernel whose mangled name contains `marker`."""
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % marker, l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return lines[start:end + 1]


def is_inst(l):
    s = l.strip()
    return bool(s) and not s.startswith((";", ".", "_Z")) and not s.endswith(":") and not re.match(r"^\.L", s)


def classify(l, epilogue=False):
    s = l.strip()
    op = s.split()[0]
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op == "s_barrier":
        return "barrier"
    if op.startswith("s_waitcnt") or op == "s_nop":
        return "wait"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        if op.startswith("v_perm_b32"):
            return "split"
        if op.startswith("v_and_b32") and ("0xffff0000" in s or "-65536" in s):
            return "split"
        if re.match(r"v_(add|sub|subrev|mul|fma|fmac|mac|max|min|pk_add|pk_mul|pk_fma)_f(32|64)", op) or op.startswith(("v_cvt_f64", "v_cvt_f32_f64", "v_rcp", "v_rsq", "v_sqrt")):
            return "fp"
        return "address"
    return "other"


def breakdown(region, epilogue=False):
    c = collections.Counter()
    for l in region:
        if is_inst(l):
            c[classify(l, epilogue)] += 1
    # the split's residual subtractions: one v_sub_f32 per v_and (8 and 8 per split3q) -- move them from "fp" to "split"
    n_and = sum(1 for l in region if is_inst(l) and l.strip().startswith("v_and_b32") and ("0xffff0000" in l or "-65536" in l))
    if not epilogue:
        mv = min(n_and, c["fp"])
        c["split"] += mv
        c["fp"] -= mv
        c["transform"] = c.pop("fp")
    else:
        c["out-xform"] = c.pop("fp", 0)
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--md", default=None)
    args = ap.parse_args()
    lines = compile_asm(os.path.join(ROOT, "dawn-pytorch_amd", "csrc", "conv3x3_wino.hip"))
    k = kernel_lines(lines, "conv3x3_wino_kernelILi0E")
    # basic blocks carry the loop they belong to in their label comments ("in Loop: Header=BB0_37 Depth=2", "Parent Loop BB0_33 Depth=1"
    # + "This Inner Loop Header: Depth=2" on the header itself): the chunk loop = the depth-2 loop, the tile loop = its depth-1 parent
    hdr2 = hdr1 = None
    for i, l in enumerate(k):
        m = re.match(r"^\.L(BB\d+_\d+):\s*;\s*Parent Loop (BB\d+_\d+) Depth=1", l)
        if m and i + 1 < len(k) and "Loop Header: Depth=2" in k[i + 1]:
            hdr2, hdr1 = m.group(1), m.group(2)
            break
    assert hdr2, "no depth-2 loop found"
    where, cur = [], "pro"
    seen_tile = False
    for i, l in enumerate(k):
        m = re.match(r"^\.L(BB\d+_\d+):(.*)$", l)
        if m:
            lab, com = m.group(1), m.group(2)
            nxt = k[i + 1] if i + 1 < len(k) else ""
            if lab == hdr2 or f"Header={hdr2} Depth=2" in com:
                cur = "chunk"
            elif lab == hdr1 or f"Header={hdr1} Depth=1" in com:
                cur, seen_tile = "tile", True
            else:
                cur = "tail" if seen_tile else "pro"
        where.append(cur)
    chunk = [l for l, w in zip(k, where) if w == "chunk"]
    bars = [i for i, l in enumerate(chunk) if l.strip() == "s_barrier"]
    assert len(bars) == 2, f"expected 2 barriers in the chunk loop, found {len(bars)}"
    phaseA, phaseB = chunk[:bars[0] + 1], chunk[bars[0] + 1:]
    epi = [l for l, w in zip(k, where) if w == "tile"]
    pro = [l for l, w in zip(k, where) if w == "pro"]
    tail = [l for l, w in zip(k, where) if w == "tail"]
    regions = [("prologue (once per workgroup)", breakdown(pro, True)), ("phase A (per chunk)", breakdown(phaseA)),
               ("phase B (per chunk)", breakdown(phaseB)), ("epilogue + tile bookkeeping (per tile)", breakdown(epi, True)),
               ("GroupNorm hand-off (once per workgroup)", breakdown(tail, True))]
    cols = ["mfma", "transform", "split", "out-xform", "address", "lds", "vmem", "salu", "wait", "barrier", "branch", "other"]
    out = []
    out.append("| region (static, per WAVE) | " + " | ".join(cols) + " | VALU total |")
    out.append("|---|" + "---|" * (len(cols) + 1))
    valu = {}
    for name, c in regions:
        v = c["transform"] + c["split"] + c["out-xform"] + c["address"]
        valu[name] = v
        out.append(f"| {name} | " + " | ".join(str(c.get(x, 0)) for x in cols) + f" | {v} |")
    out.append("")
    out.append("(the chunk loop's two phases are fully unrolled straight-line code: static counts = instructions a wave issues per phase; "
               "the epilogue region contains two data-dependent branches -- residual / bias present or not -- so its count is an upper bound "
               "by a few instructions)")
    out.append("")
    out.append("| launch shape | tiles | chunks per tile | predicted SQ_INSTS_VALU (wave-instructions) | of which main loop | epilogue |")
    out.append("|---|---|---|---|---|---|")
    for name, M, N, Cin in (("level 0, K = 576 (M = 819,200, N = 64)", 819200, 64, 64), ("level 0, K = 1152", 819200, 64, 128),
                            ("level 1, N = 128, K = 1152 (M = 204,800)", 204800, 128, 128), ("level 3, N = 512, K = 4608 (M = 12,800)", 12800, 512, 512)):
        tiles = M // 256 * (N // 64)
        nC = Cin // 16
        wg = min(tiles, 256)
        main_ = tiles * nC * 8 * (valu["phase A (per chunk)"] + valu["phase B (per chunk)"])
        ep = tiles * 8 * valu["epilogue + tile bookkeeping (per tile)"]
        once = wg * 8 * (valu["prologue (once per workgroup)"] + valu["GroupNorm hand-off (once per workgroup)"])
        out.append(f"| {name} | {tiles} | {nC} | {(main_ + ep + once) / 1e6:.1f} M | {main_ / 1e6:.1f} M | {ep / 1e6:.1f} M |")
    text = "\n".join(out)
    print(text)
    if args.md:
        open(args.md, "w").write(text + "\n")


if __name__ == "__main__":
    main()
